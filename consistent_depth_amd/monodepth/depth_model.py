"""Plugin base class for depth networks (mirrors /root/reference/monodepth/depth_model.py:8-38).

Contract kept from the reference: `forward(images[..., C, H, W], metadata=None)` returns
positive *depth* `[..., H, W]`; subclasses implement `estimate_depth(images)` and
`save(file_name)`; callers also rely on a zero-argument constructor, the class attributes
`align`, `learning_rate`, `lambda_view_baseline` (params.py:110-119) and
`train()/eval()/parameters()` (depth_fine_tuning.py:182,224,233,241).

Extension used by the fused engine: a model may expose `estimate_raw(images)` plus a class
attribute `depth_mode` (consistent_depth_amd.loss.consistency_loss.DEPTH_*) so the
exp / reciprocal head is fused into the loss kernel instead of being a separate pass.
"""
from abc import ABC, abstractmethod

import torch

from ..loss.consistency_loss import DEPTH_IDENTITY


class DepthModel(torch.nn.Module, ABC):
    align = 1
    learning_rate = 0.0
    lambda_view_baseline = 0.0
    depth_mode = DEPTH_IDENTITY

    def __init__(self):
        super().__init__()

    def forward(self, images, metadata=None):
        depth = self.estimate_depth(images)
        # optional per-frame post-scale, metadata["scales"]: (..., N, 1) -> broadcast over H, W
        if metadata is not None and "scales" in metadata:
            depth = depth * metadata["scales"].unsqueeze(3).to(depth.device)
        return depth

    @abstractmethod
    def estimate_depth(self, images) -> torch.Tensor:
        ...

    def estimate_raw(self, images) -> torch.Tensor:
        """Network output before the depth head (see `depth_mode`); default: the depth itself."""
        return self.estimate_depth(images)

    @abstractmethod
    def save(self, file_name):
        ...
