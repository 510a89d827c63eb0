"""Sum of the parameter-regularisation and geometric-consistency losses
(mirrors /root/reference/loss/joint_loss.py:15-47: same constructor, call and outputs --
`loss` of shape (1,), dict of per-pair sub-losses)."""
from __future__ import annotations

from typing import List, Optional

import torch

from .consistency_loss import DEPTH_IDENTITY, ConsistencyLoss
from .parameter_loss import ParameterLoss


class JointLoss(torch.nn.Module):
    def __init__(self, opt, parameters_init=None, depth_mode: int = DEPTH_IDENTITY):
        super().__init__()
        self.opt = opt
        if opt.lambda_parameter > 0:
            assert parameters_init is not None
            self.parameter_loss = ParameterLoss(parameters_init, opt)
        if opt.lambda_view_baseline > 0 or opt.lambda_reprojection > 0:
            self.consistency_loss = ConsistencyLoss(opt, depth_mode=depth_mode)

    def __call__(self, depths, metadata, parameters: Optional[List[torch.nn.Parameter]] = None):
        loss = torch.zeros(1, dtype=torch.float32, device=depths.device)
        batch_losses = {}
        if self.opt.lambda_parameter > 0:
            assert parameters is not None
            p_loss, p_parts = self.parameter_loss(list(parameters))
            loss = loss + p_loss
            batch_losses.update(p_parts)
        if self.opt.lambda_view_baseline > 0 or self.opt.lambda_reprojection > 0:
            c_loss, c_parts = self.consistency_loss(depths, metadata)
            loss = loss + c_loss
            batch_losses.update(c_parts)
        return loss, batch_losses
