"""One fine-tuning step of the hot path (reference loop body: depth_fine_tuning.py:264-293).

    raw   = model.estimate_raw(images)          # CNN forward (hourglass / MiDaS)
    loss  = JointLoss(raw, metadata)            # fused HIP loss, depth head fused in
    loss.backward()                             # CNN backward
    [all-reduce flat grads + loss scalar]       # data parallel only, one RCCL call
    FlatAdam.step(guard_loss=loss)              # one HIP launch; NaN guard on the device

Compared with the reference's body there is no per-step host synchronisation: the reference
prints `loss[0]` (a D2H sync) and tests `torch.isnan(loss)` on the host every step (:275-280);
here the NaN-skip is evaluated inside the Adam kernel and losses are fetched only when logged.
"""
from __future__ import annotations

import torch

from . import optimizer as cd_optimizer
from . import parallel
from .loss.joint_loss import JointLoss


class FineTuneStep:
    def __init__(self, model, params, world: int = None):
        """`params` needs lambda_reprojection / lambda_view_baseline / lambda_parameter /
        learning_rate / optimizer (as produced by consistent_depth_amd.params)."""
        self.model = model
        self.world = world if world is not None else parallel.world_size()
        plist = list(model.parameters())
        init = [p.detach().clone() for p in plist] if params.lambda_parameter > 0 else None
        self.criterion = JointLoss(params, parameters_init=init, depth_mode=model.depth_mode)
        self.opt = cd_optimizer.create(getattr(params, "optimizer", "Adam"), plist, params.learning_rate,
                                       betas=(0.9, 0.999))
        self._plist = plist

    def forward_loss(self, images, metadata):
        raw = self.model.estimate_raw(images)
        return self.criterion(raw, metadata, parameters=self._plist)

    def __call__(self, images, metadata):
        """Runs one optimisation step; returns (loss (1,), {name: (B,)}) as device tensors."""
        raw = self.model.estimate_raw(images)
        self.opt.zero_grad()
        loss, parts = self.criterion(raw, metadata, parameters=self._plist)
        loss.backward()
        guard = loss.detach()
        if self.world > 1:
            self.opt.loss_slot.copy_(guard.reshape(1))
            parallel.allreduce_sum_(self.opt.reduce_buffer)
            guard = self.opt.loss_slot
        self.opt.step(grad_scale=1.0 / self.world, guard_loss=guard)
        return loss.detach(), parts

    @torch.no_grad()
    def evaluate(self, images, metadata):
        """Validation forward: model stays in whatever mode it is in (the reference keeps
        train-mode BatchNorm during validation, depth_fine_tuning.py:241,327-328)."""
        raw = self.model.estimate_raw(images)
        loss, parts = self.criterion(raw, metadata, parameters=self._plist)
        return raw, loss, parts
