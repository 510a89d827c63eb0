"""L1 pull of the network weights towards their initial values
(mirrors /root/reference/loss/parameter_loss.py:7-19; off by default, lambda_parameter = 0).

The reference concatenates |p - p0| of every tensor on every step; here the parameters live
in one flat buffer (consistent_depth_amd.optimizer.FlatParams) and a single fused reduction
(cd_l1_distance) computes the sum; the gradient lambda * sign(p - p0) is one elementwise op.
"""
from __future__ import annotations

import torch

from .. import _native


class _L1Distance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, flat_init):
        lib = _native.lib()
        n = flat.numel()
        ws_bytes = lib.cd_l1_distance_workspace_bytes(n)
        ws = _native.workspace("l1_distance", ws_bytes, flat.device)
        out = torch.empty(1, dtype=torch.float32, device=flat.device)
        rc = lib.cd_l1_distance(_native.dev_ptr(flat, "params"), _native.dev_ptr(flat_init, "params_init"), n,
                                out.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr(flat.device))
        _native.check(rc, "cd_l1_distance")
        ctx.save_for_backward(flat, flat_init)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        flat, flat_init = ctx.saved_tensors
        return torch.sign(flat - flat_init) * g, None


class ParameterLoss(torch.nn.Module):
    def __init__(self, parameters_init, opt):
        super().__init__()
        assert opt.lambda_parameter > 0
        self.opt = opt
        self.parameters_init = [p.detach() for p in parameters_init]
        self._flat_init = torch.cat([p.reshape(-1) for p in self.parameters_init]).contiguous()

    def __call__(self, parameters):
        flat = torch.cat([p.reshape(-1) for p in parameters]).contiguous()
        loss = self.opt.lambda_parameter * _L1Distance.apply(flat, self._flat_init)
        return loss, {"parameter_loss": loss.reshape(1, -1)}
