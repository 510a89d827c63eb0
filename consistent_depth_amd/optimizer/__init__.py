"""Optimiser factory (mirrors /root/reference/optimizer/__init__.py:5-17: name -> class map,
`create(name, params, lr, **kw)`); "Adam" resolves to the flat-buffer HIP Adam below.

FlatAdam re-homes every parameter (and its .grad) as a view into ONE contiguous fp32 buffer:
  * the Adam step is a single launch of cd_adam_step_flat over the buffer (the reference's
    torch.optim.Adam issues several launches per parameter tensor, ~316 tensors for `mc`);
  * zero_grad is one memset;
  * data-parallel training all-reduces the same flat gradient buffer in one RCCL call
    (consistent_depth_amd.parallel).
Parameters whose grad stays zero (the unused `uncertainty_layer` of `mc`,
mannequin_challenge_model.py:60) are left untouched, like torch skipping grad=None.
"""
from __future__ import annotations

from typing import Iterable

import torch

from .. import _native

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary inside the flat buffer


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params]
        if not params:
            raise ValueError("FlatAdam: empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs parameters on the HIP device (no CPU path)")
        offs, n = [], 0
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("FlatAdam: all parameters must be float32 on one device")
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        # one spare 256-byte slot behind the gradients carries the step's loss scalar, so a
        # data-parallel step all-reduces gradients AND the NaN-guard scalar in ONE collective
        self._grad_store = torch.zeros(n + _ALIGN, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_store[:n]
        self.loss_slot = self._grad_store[n:n + 1]
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self._params, self._offsets = params, offs
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.step_count = 0  # host-side count (un-guarded steps)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)  # device-side count (guarded steps)
        self.numel = sum(p.numel() for p in params)

    @torch.no_grad()
    def load_moments(self, exp_avg, exp_avg_sq, steps: int):
        """Resume: the Adam moments (one tensor per parameter, in the order of the constructor's parameter list) and the number of
        steps already taken.  (torch.optim.Adam keeps the same three things per parameter: `exp_avg`, `exp_avg_sq`, `step`.)"""
        if len(exp_avg) != len(self._params) or len(exp_avg_sq) != len(self._params):
            raise ValueError("FlatAdam.load_moments: one moment tensor per parameter")
        for p, o, m1, m2 in zip(self._params, self._offsets, exp_avg, exp_avg_sq):
            if m1.numel() != p.numel() or m2.numel() != p.numel():
                raise ValueError("FlatAdam.load_moments: moment shape does not match its parameter")
            self.exp_avg[o:o + p.numel()].copy_(m1.reshape(-1).to(self.exp_avg))
            self.exp_avg_sq[o:o + p.numel()].copy_(m2.reshape(-1).to(self.exp_avg_sq))
        self.step_count = int(steps)
        self.step_dev.fill_(int(steps))

    def zero_grad(self, set_to_none: bool = False):
        # grads are views into the flat buffer: keep them bound, clear with one memset
        _native.zero_(self._grad_store)
        for p, o in zip(self._params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    @property
    def reduce_buffer(self) -> torch.Tensor:
        """Gradients + loss slot: the tensor a data-parallel step all-reduces (sum)."""
        return self._grad_store

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, guard_loss: torch.Tensor = None):
        """One Adam step.  With `guard_loss` (a device scalar) the step is skipped on the device
        when the loss is NaN and the step counter lives on the device: no host sync at all."""
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        if guard_loss is not None:
            rc = _native.lib().cd_adam_step_flat_guarded(
                _native.dev_ptr(self.flat_param), _native.dev_ptr(self.flat_grad), _native.dev_ptr(self.exp_avg),
                _native.dev_ptr(self.exp_avg_sq), self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), self.step_dev.data_ptr(),
                _native.dev_ptr(guard_loss.reshape(-1), "guard_loss"), float(grad_scale),
                _native.stream_ptr(self.flat_param.device))
            _native.check(rc, "cd_adam_step_flat_guarded")
            return loss
        self.step_count += 1
        rc = _native.lib().cd_adam_step_flat(
            _native.dev_ptr(self.flat_param), _native.dev_ptr(self.flat_grad), _native.dev_ptr(self.exp_avg),
            _native.dev_ptr(self.exp_avg_sq), self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
            float(g["betas"][1]), float(g["eps"]), self.step_count, float(grad_scale),
            _native.stream_ptr(self.flat_param.device))
        _native.check(rc, "cd_adam_step_flat")
        return loss


OPTIMIZER_MAP = {"Adam": FlatAdam}
OPTIMIZER_NAMES = OPTIMIZER_MAP.keys()
OPTIMIZER_CLASSES = OPTIMIZER_MAP.values()


def create(optimizer_name: str, *args, **kwargs):
    return OPTIMIZER_MAP[optimizer_name](*args, **kwargs)
