"""TEST INFRASTRUCTURE ONLY -- the reference's fine-tuning step restated on the CPU
(bench.py `cpu_baseline` leg, kind "port"; tests use it as the step-level reference).

One step = depth_fine_tuning.py:264-283 of the reference: hourglass forward (oracle/hourglass_ref,
train-mode BN) -> depth = exp(pred) -> geometric-consistency loss + d loss/d depth (plain-C oracle,
cd_oracle.c) -> backward through the CNN (torch autograd, all host threads) -> torch.optim.Adam.
Nothing here is reachable from consistent_depth_amd/.
"""
import time

import numpy as np
import torch

from . import hourglass_ref, oracle


class CpuFineTuner:
    def __init__(self, state_dict, lr=4e-4, lambda_r=1.0, lambda_b=0.1, dtype=torch.float32, device="cpu"):
        """`device`: where torch evaluates the hourglass and Adam (the loss stays the plain-C oracle on the host).  "cpu" is the
        reference path; "cuda" runs the SAME torch program (ATen's own fp64 kernels, nothing of consistent_depth_amd) on the GPU
        -- used only to produce long fp64 ground-truth runs in minutes instead of hours (oracle/gen_golden_loop_384.py, where it is
        cross-checked against the CPU run epoch by epoch)."""
        self.dtype, self.device = dtype, torch.device(device)
        self.state = {k: v.detach().clone().to(self.device, dtype if v.is_floating_point() else v.dtype)
                      for k, v in state_dict.items()}
        self.param_keys = [k for k in self.state if k.endswith(".weight") or k.endswith(".bias")]
        for k in self.param_keys:
            self.state[k].requires_grad_(True)
        self.opt = torch.optim.Adam([self.state[k] for k in self.param_keys], lr, betas=(0.9, 0.999))
        self.lambda_r, self.lambda_b = lambda_r, lambda_b

    def set_adam_state(self, exp_avg, exp_avg_sq, step):
        """Continue from a warm optimiser: per-parameter first / second moments (dicts keyed like the state dict; a missing
        key = zeros) and the number of steps already taken -- exactly torch.optim.Adam's own state layout."""
        for k in self.param_keys:
            p = self.state[k]
            m = exp_avg.get(k)
            v = exp_avg_sq.get(k)
            self.opt.state[p] = {
                "step": torch.tensor(float(step)),
                "exp_avg": (torch.zeros_like(p) if m is None else m.detach().to(self.device, self.dtype).reshape(p.shape).clone()),
                "exp_avg_sq": (torch.zeros_like(p) if v is None else v.detach().to(self.device, self.dtype).reshape(p.shape).clone()),
            }

    def step(self, images, batch):
        """images (B,2,3,H,W) numpy/tensor; batch: dict with flows/masks/intrinsics/extrinsics (numpy)."""
        np_dtype = np.float64 if self.dtype == torch.float64 else np.float32
        x = torch.as_tensor(images, dtype=self.dtype).reshape((-1,) + tuple(images.shape[-3:])).to(self.device)
        pred, _ = hourglass_ref.forward(self.state, x, training=True, update_running_stats=True)
        B = images.shape[0]
        depth = torch.exp(pred).reshape(B, 2, *pred.shape[-2:])
        out = oracle.consistency_loss(depth.detach().cpu().numpy(), batch["flows"], batch["masks"], batch["intrinsics"],
                                      batch["extrinsics"], self.lambda_r, self.lambda_b, dtype=np_dtype)
        self.opt.zero_grad()
        if not np.isnan(out["total"][0]):
            depth.backward(torch.as_tensor(out["grad_depth"]).to(self.device))
            self.opt.step()
        return out, depth.detach()


def time_steps(state_dict, images, batch, n_steps=5, warmup=1, threads=None, loss_steps=10):
    """Seconds per full step and per loss-only evaluation (loss + d loss / d depth of the same batch) -- MEDIANS over the timed
    repetitions (>= 3 full + >= 10 loss-only steps).  Returns (full_s, loss_s, all full-step times)."""
    if threads:
        torch.set_num_threads(threads)
    ft = CpuFineTuner(state_dict)
    depth = None
    for _ in range(warmup):
        _, depth = ft.step(images, batch)
    full = []
    for _ in range(n_steps):
        t0 = time.perf_counter()
        _, depth = ft.step(images, batch)
        full.append(time.perf_counter() - t0)
    d = depth.cpu().numpy()
    loss = []
    for i in range(loss_steps + 1):
        t0 = time.perf_counter()
        oracle.consistency_loss(d, batch["flows"], batch["masks"], batch["intrinsics"], batch["extrinsics"], ft.lambda_r, ft.lambda_b,
                                dtype=np.float32)
        if i:            # (the first evaluation is the warm-up)
            loss.append(time.perf_counter() - t0)
    return float(np.median(full)), float(np.median(loss)) if loss else float("nan"), full
