"""ctypes face of the C hourglass engine (cd_hourglass_*, csrc/hourglass.hip): the whole CNN -- plan, buffers, forward,
explicit backward -- behind one handle of the C ABI, for hosts without Python (INTEGRATION.md section 2 drives a full
fine-tuning step with it).  Inside this package the Python orchestration (hourglass_engine.py: side streams, timed launch
shapes, autograd node) stays the default execution path; this wrapper exists to exercise and test the handle API."""
from __future__ import annotations

import ctypes

import torch

from .. import _native


class CHourglass:
    def __init__(self, N: int, H: int, W: int):
        self._lib = _native.lib()
        h = ctypes.c_void_p()
        _native.check(self._lib.cd_hourglass_create(N, H, W, ctypes.byref(h)), "cd_hourglass_create")
        self._h, self.shape = h, (N, H, W)
        self.n_param, self.n_bn = self._lib.cd_hourglass_param_floats(h), self._lib.cd_hourglass_bn_floats(h)

    def close(self):
        if self._h:
            self._lib.cd_hourglass_destroy(self._h)
            self._h = None

    __del__ = close

    def param_layout(self):
        """[(offset, shape)] of every parameter tensor, in named_parameters() order."""
        out = []
        for i in range(self._lib.cd_hourglass_param_count(self._h)):
            off, shp = ctypes.c_size_t(0), (ctypes.c_int * 4)()
            _native.check(self._lib.cd_hourglass_param_info(self._h, i, ctypes.byref(off), shp), "cd_hourglass_param_info")
            out.append((off.value, tuple(s for s in shp if s > 0)))
        return out

    def load_module(self, net: torch.nn.Module):
        """Copy an HourglassModel's parameters and BatchNorm running statistics into the engine."""
        flat = torch.zeros(self.n_param, dtype=torch.float32)
        layout = self.param_layout()
        named = list(net.named_parameters())
        assert len(named) == len(layout)
        for (name, p), (off, shp) in zip(named, layout):
            assert tuple(p.shape) == shp, (name, tuple(p.shape), shp)
            flat[off:off + p.numel()] = p.detach().float().cpu().reshape(-1)
        bn = torch.cat([torch.cat([m.running_mean.detach().float().cpu(), m.running_var.detach().float().cpu()])
                        for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)])
        assert bn.numel() == self.n_bn
        _native.check(self._lib.cd_hourglass_load_state(self._h, flat.data_ptr(), bn.data_ptr(), None), "cd_hourglass_load_state")
        torch.cuda.synchronize()

    def state(self):
        """(flat parameters, flat running statistics) as CPU tensors."""
        flat, bn = torch.empty(self.n_param), torch.empty(self.n_bn)
        _native.check(self._lib.cd_hourglass_save_state(self._h, flat.data_ptr(), bn.data_ptr(), None), "cd_hourglass_save_state")
        torch.cuda.synchronize()
        return flat, bn

    def grads(self) -> torch.Tensor:
        """The flat gradient buffer (a copy)."""
        out = torch.empty(self.n_param, dtype=torch.float32, device="cuda")
        _native.check(self._lib.cd_copy_f32(self._lib.cd_hourglass_grads(self._h), out.data_ptr(), self.n_param,
                                            _native.stream_ptr()), "cd_copy_f32")
        torch.cuda.synchronize()
        return out

    def zero_grad(self):
        _native.check(self._lib.cd_hourglass_zero_grad(self._h, _native.stream_ptr()), "cd_hourglass_zero_grad")

    def forward(self, images: torch.Tensor, training: bool = True) -> torch.Tensor:
        N, H, W = self.shape
        assert tuple(images.shape) == (N, 3, H, W)
        pred = torch.empty(N, 1, H, W, dtype=torch.float32, device=images.device)
        _native.check(self._lib.cd_hourglass_forward(self._h, _native.dev_ptr(images, "images"), pred.data_ptr(), int(training),
                                                     _native.stream_ptr(images.device)), "cd_hourglass_forward")
        return pred

    def backward(self, dpred: torch.Tensor):
        _native.check(self._lib.cd_hourglass_backward(self._h, _native.dev_ptr(dpred.contiguous(), "dpred"),
                                                      _native.stream_ptr(dpred.device)), "cd_hourglass_backward")
