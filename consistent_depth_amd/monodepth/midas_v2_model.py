"""`midas2` depth model plugin (reference: /root/reference/monodepth/midas_v2_model.py:12-73).

Class attributes (:14-16), ImageNet normalisation (:45-48,58-59) and the disparity->depth
reciprocal (:67) follow the reference.  The MidasNet source is an un-vendored submodule
(.gitmodules:7-9) and its weights are unreachable; the backbone is restated in
consistent_depth_amd/monodepth/midas_net.py (ResNeXt-101 32x8d encoder + 4 feature-fusion
blocks, random init) -- "parity unpinned" for real checkpoints.
"""
from __future__ import annotations

import os

import torch

from ..loss.consistency_loss import DEPTH_RECIPROCAL
from .depth_model import DepthModel


class MidasV2Model(DepthModel):
    align = 32
    learning_rate = 0.0001
    lambda_view_baseline = 0.0001
    depth_mode = DEPTH_RECIPROCAL  # depth = 1 / disparity, :67

    RANDOM_INIT_DISPARITY = 300.0

    def __init__(self, support_cpu: bool = False, pretrained: bool = False, seed: int = 0, backend: str = None):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("MidasV2Model needs the HIP device (no CPU path in consistent_depth_amd)")
        from .midas_net import MidasNet
        self.device = torch.device("cuda", torch.cuda.current_device())
        st = torch.random.get_rng_state()
        torch.manual_seed(seed)
        # convolutions: "hip" = hand-written gfx950 MFMA kernels (ops/conv_layer.py), "torch" = PyTorch-ROCm / MIOpen
        self.backend = backend or os.environ.get("CD_AMD_MIDAS_BACKEND", "hip")
        self.model = MidasNet(non_negative=True, backend=self.backend)
        torch.random.set_rng_state(st)
        weights = os.environ.get("CD_AMD_MIDAS_WEIGHTS", os.path.join("checkpoints", "midas_v2.pt"))
        self.pretrained = os.path.isfile(weights)
        if self.pretrained:
            self.model.load_state_dict(torch.load(weights, map_location="cpu"))
        else:
            # random-init stand-in (no network for model-f46da743.pt): the predicted inverse depth starts at the magnitude of
            # real MiDaS-v2 outputs (1e2..1e3) instead of straddling zero.  The final ReLU would otherwise zero about half the
            # pixels (depth = 1/0), and with an O(1) output the first Adam steps (+-lr on each of 105 M weights, no
            # normalisation layer in the decoder) move the prediction by ~0.7 per step and push pixels through zero within a
            # handful of steps: a NaN loss that the guard then skips forever (measured: tools/exp/diag_midas_nan.py)
            with torch.no_grad():
                head = self.model.scratch.output_conv[4]
                head.weight.mul_(0.1)
                head.bias.fill_(self.RANDOM_INIT_DISPARITY)
        self.model.to(self.device)
        self.register_buffer("norm_mean", torch.tensor([0.485, 0.456, 0.406]).reshape(1, -1, 1, 1).to(self.device))
        self.register_buffer("norm_stdev", torch.tensor([0.229, 0.224, 0.225]).reshape(1, -1, 1, 1).to(self.device))

    def estimate_raw(self, images):
        shape = images.shape
        C, H, W = shape[-3:]
        x = images.reshape(-1, C, H, W).to(self.device)
        x = (x - self.norm_mean) / self.norm_stdev
        out = self.model(x)
        return out.reshape(shape[:-3] + out.shape[-2:])

    def estimate_depth(self, images):
        return self.estimate_raw(images).reciprocal()

    def weights_updated(self):
        """Called by the fine-tuning step after every optimiser update: the packed copies of the filters are stale."""
        pool = getattr(self.model, "_pack_pool", None)
        if pool is not None:
            pool.invalidate()

    def save(self, file_name):
        torch.save(self.model.state_dict(), file_name)
