"""The training objective of the fine-tuning step: every active loss term, summed.

Same operator surface as the reference's JointLoss (/root/reference/loss/joint_loss.py:15-47): constructed from the
option namespace (+ the initial parameters when the parameter regulariser is on), called with (depths, metadata,
parameters), returns `(loss (1,), {sub-loss name: per-pair values})`.  Here the terms are resolved ONCE at construction
into a list of callables, so the per-step call is a plain loop without option tests; `depth_mode` lets a model hand over
its raw network output (the depth head is fused into the consistency kernel)."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

from .consistency_loss import DEPTH_IDENTITY, ConsistencyLoss
from .parameter_loss import ParameterLoss

Term = Callable[[torch.Tensor, dict, Optional[Iterable[torch.nn.Parameter]]], Tuple[torch.Tensor, Dict[str, torch.Tensor]]]


class JointLoss(torch.nn.Module):
    def __init__(self, opt, parameters_init=None, depth_mode: int = DEPTH_IDENTITY):
        super().__init__()
        self.opt = opt
        self._terms: List[Term] = []
        if opt.lambda_parameter > 0:
            if parameters_init is None:
                raise AssertionError("lambda_parameter > 0 needs the initial parameters")   # the reference asserts
            self.parameter_loss = ParameterLoss(parameters_init, opt)
            self._terms.append(self._parameter_term)
        if max(opt.lambda_view_baseline, opt.lambda_reprojection) > 0:
            self.consistency_loss = ConsistencyLoss(opt, depth_mode=depth_mode)
            self._terms.append(lambda depths, metadata, _params: self.consistency_loss(depths, metadata))

    def _parameter_term(self, _depths, _metadata, parameters):
        if parameters is None:
            raise AssertionError("the parameter regulariser needs the current parameters")
        return self.parameter_loss(list(parameters))

    def __call__(self, depths, metadata, parameters=None):
        if len(self._terms) == 1:       # the default configuration (lambda_parameter = 0): the term's value IS the loss -- no zeros(1), add and
            value, parts = self._terms[0](depths, metadata, parameters)       # broadcast-sum in backward (three framework launches per step)
            return value.reshape(1), dict(parts)
        total = torch.zeros(1, dtype=torch.float32, device=depths.device)
        per_pair: Dict[str, torch.Tensor] = {}
        for term in self._terms:
            value, parts = term(depths, metadata, parameters)
            total = total + value
            per_pair.update(parts)
        return total, per_pair
