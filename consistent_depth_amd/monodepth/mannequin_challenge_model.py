"""`mc` depth model: Mannequin-Challenge hourglass behind the DepthModel plugin surface.

Mirrors /root/reference/monodepth/mannequin_challenge_model.py:15-73 (class attributes
:17-19, train/eval/parameters :43-50, estimate_depth :52-69, save :71-73).

Differences, all forced by the environment and stated here:
  * the pretrained checkpoint URL (:30) is unreachable (no network): weights are loaded from
    $CD_AMD_MC_WEIGHTS or <cwd>/checkpoints/mc.pth when present, otherwise the network is
    randomly initialised from a fixed seed (BASELINE config 3: "random-init mc hourglass");
  * two execution back ends for the same parameters:
      backend="torch": convolutions through PyTorch-ROCm/MIOpen (BASELINE config 2),
      backend="hip":   the hand-written gfx950 hourglass engine (BASELINE config 3);
    selected by the constructor argument or $CD_AMD_MC_BACKEND (default "hip": every layer of the
    hourglass runs on the hand-written engine, forward and backward).
"""
from __future__ import annotations

import os

import torch

from ..loss.consistency_loss import DEPTH_EXP
from .depth_model import DepthModel
from .hourglass import HourglassModel, load_state_dict_any_prefix

DEFAULT_SEED = 0


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("MannequinChallengeModel needs the HIP device (no CPU path in consistent_depth_amd)")
    return torch.device("cuda", torch.cuda.current_device())


class MannequinChallengeModel(DepthModel):
    # Requirements and default settings (mannequin_challenge_model.py:17-19)
    align = 16
    learning_rate = 0.0004
    lambda_view_baseline = 0.1
    depth_mode = DEPTH_EXP  # depth = exp(pred_d), :66

    def __init__(self, backend: str = None, seed: int = DEFAULT_SEED, device=None):
        super().__init__()
        self.backend = backend or os.environ.get("CD_AMD_MC_BACKEND", "hip")
        if self.backend not in ("torch", "hip"):
            raise ValueError(f"unknown mc backend '{self.backend}'")
        self.device = torch.device(device) if device is not None else _device()
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.netG = HourglassModel(3)
        torch.random.set_rng_state(gen_state)
        weights = os.environ.get("CD_AMD_MC_WEIGHTS", os.path.join("checkpoints", "mc.pth"))
        self.pretrained = os.path.isfile(weights)
        if self.pretrained:
            load_state_dict_any_prefix(self.netG, torch.load(weights, map_location="cpu"))
        self.netG.to(self.device)
        self._engine = None
        if self.backend == "hip":
            from .hourglass_engine import HourglassEngine  # raises if the native library is missing
            self._engine = HourglassEngine(self.netG)

    # the reference forwards train/eval/parameters to netG (:43-50)
    def train(self, mode: bool = True):
        self.netG.train(mode)
        return self

    def eval(self):
        self.netG.eval()
        return self

    def parameters(self, recurse: bool = True):
        return self.netG.parameters()

    def estimate_raw(self, images):
        """(..., 3, H, W) RGB in [0,1] -> log-depth (..., H, W)   (:52-64 without the exp)."""
        images = images.to(self.device, non_blocking=True)
        shape = images.shape
        C, H, W = shape[-3:]
        x = images.reshape(-1, C, H, W)
        if self._engine is not None:
            pred = self._engine.forward(x)
        else:
            pred, _ = self.netG(x)
        return pred.reshape(shape[:-3] + pred.shape[-2:])

    def estimate_depth(self, images):
        return torch.exp(self.estimate_raw(images))

    def save(self, file_name):
        torch.save(self.netG.state_dict(), file_name)
