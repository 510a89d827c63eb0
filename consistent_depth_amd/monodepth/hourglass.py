"""Mannequin-Challenge hourglass depth CNN -- architecture table + PyTorch container.

The network source is NOT in /root/reference (un-vendored submodule
monodepth/mannequin_challenge, .gitmodules:4-6; call sites
monodepth/mannequin_challenge_model.py:10,34-41,60).  The architecture below is restated
from the published google/mannequinchallenge `models/hourglass.py` as summarised in
SURVEY.md appendix A.3 and is corroborated only by the parameter count (5 357 730) and the
(pred_d, confidence) return signature -- "parity unpinned" for real checkpoints.

This module is (a) the parameter container whose state_dict keys follow the upstream
checkpoint layout (`seq.0.weight`, `seq.3.list.0.1.convs.2.3.weight`, `pred_layer.weight`,
optionally `module.`-prefixed), so a real mc.pth loads by key, and (b) the
"convs on PyTorch-ROCm (MIOpen)" execution path of BASELINE config 2.  The hand-written
HIP execution path walks the same `plan()` (consistent_depth_amd/monodepth/hourglass_engine.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

# inception(C_in, [[a], [k1, mid1, out1], [k2, mid2, out2], [k3, mid3, out3]])
INCEPTION = {
    "A":  (128, [[16], [3, 32, 16], [7, 32, 16], [11, 32, 16]]),
    "A2": (128, [[16], [3, 64, 16], [7, 64, 16], [11, 64, 16]]),
    "B":  (128, [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]]),
    "B2": (128, [[32], [3, 64, 32], [5, 64, 32], [7, 64, 32]]),
    "C":  (128, [[32], [3, 64, 32], [7, 64, 32], [11, 64, 32]]),
    "D":  (128, [[64], [3, 32, 64], [5, 32, 64], [7, 32, 64]]),
    "E":  (256, [[64], [3, 32, 64], [5, 32, 64], [7, 32, 64]]),
    "F":  (256, [[64], [3, 64, 64], [7, 64, 64], [11, 64, 64]]),
    "G":  (256, [[32], [3, 32, 32], [5, 32, 32], [7, 32, 32]]),
}

# ChannelsN(x) = list[0](x) + list[1](x); "pool"/"up" = AvgPool2d(2) / bilinear x2 (align_corners=True)
CHANNELS = {
    1: (["E", "E"], ["pool", "E", "E", "E", "up"]),
    2: (["E", "F"], ["pool", "E", "E", ("channels", 1), "E", "F", "up"]),
    3: (["pool", "B", "D", ("channels", 2), "E", "G", "up"], ["B", "C"]),
    4: (["pool", "B", "B", ("channels", 3), "B2", "A", "up"], ["A2"]),
}

ALIGN = 16  # H, W must be multiples of 2**4 (mannequin_challenge_model.py:17)


class Inception(nn.Module):
    def __init__(self, name: str):
        super().__init__()
        c_in, cfg = INCEPTION[name]
        self.kind = name
        self.convs = nn.ModuleList()
        self.convs.append(nn.Sequential(nn.Conv2d(c_in, cfg[0][0], 1), nn.BatchNorm2d(cfg[0][0], affine=False),
                                        nn.ReLU(True)))
        for k, mid, out in cfg[1:]:
            self.convs.append(nn.Sequential(
                nn.Conv2d(c_in, mid, 1), nn.BatchNorm2d(mid, affine=False), nn.ReLU(True),
                nn.Conv2d(mid, out, k, padding=(k - 1) // 2), nn.BatchNorm2d(out, affine=False), nn.ReLU(True)))

    def forward(self, x):
        return torch.cat([branch(x) for branch in self.convs], dim=1)


def _make(item):
    if item == "pool":
        return nn.AvgPool2d(2)
    if item == "up":
        return nn.UpsamplingBilinear2d(scale_factor=2)
    if isinstance(item, tuple):
        return Channels(item[1])
    return Inception(item)


class Channels(nn.Module):
    def __init__(self, level: int):
        super().__init__()
        self.level = level
        self.list = nn.ModuleList(nn.Sequential(*[_make(it) for it in side]) for side in CHANNELS[level])

    def forward(self, x):
        return self.list[0](x) + self.list[1](x)


class HourglassModel(nn.Module):
    """forward(images (X,3,H,W)) -> (pred_d (X,1,H,W) log-depth, pred_confidence (X,1,H,W))."""

    def __init__(self, num_input: int = 3):
        super().__init__()
        self.seq = nn.Sequential(nn.Conv2d(num_input, 128, 7, padding=3), nn.BatchNorm2d(128), nn.ReLU(True),
                                 Channels(4))
        self.uncertainty_layer = nn.Sequential(nn.Conv2d(64, 1, 3, padding=1), nn.Sigmoid())
        self.pred_layer = nn.Conv2d(64, 1, 3, padding=1)

    def forward(self, x):
        feat = self.seq(x)
        return self.pred_layer(feat), self.uncertainty_layer(feat)


def load_state_dict_any_prefix(net: nn.Module, state: dict) -> None:
    """Accept checkpoints saved from a DataParallel wrapper (`module.` prefix) or not."""
    if any(k.startswith("module.") for k in state):
        state = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state.items()}
    net.load_state_dict(state)


def count_parameters(net: nn.Module) -> int:
    return sum(p.numel() for p in net.parameters())
