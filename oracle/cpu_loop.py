"""TEST INFRASTRUCTURE ONLY -- the reference's fine-tuning LOOP restated on the CPU: `fine_tune`
(/root/reference/depth_fine_tuning.py:201-310), `eval_and_save` (:312-406) and `save_depth` (:164-199) over
oracle/cpu_step.py's step (hourglass_ref + the plain-C loss oracle + torch.optim.Adam), fp32 or fp64.

PINNED to the reference's own loop: tests/test_reference_loop_live_cpu.py runs /root/reference's `DepthFineTuner`
itself (oracle/ref_loop.py, build container only) on the same on-disk clip, initial weights and batch order in fp64
and compares every artefact of the run -- per-step losses, `eval/loss_e*.json`, `eval/depth_*.raw`, the
checkpoint, `depth/frame_*.raw` -- with this restatement's.  The GPU tests then use THIS loop (it runs on the GPU
box, the reference does not exist there) as the run-level reference of the product's DepthFineTuner.

Loop semantics restated (the things a step-level comparison cannot see):
  * validation before epoch 0 and after every `val_epoch_freq` epochs, sequential batches of `batch_size`, the last
    one short (:215-218, DataLoader drop_last=False);
  * validation runs the network in TRAIN mode under no_grad (:241 sets train() once; :327-328): batch statistics
    normalise, and the running statistics ARE updated by every validation batch -- they only matter for the
    eval-mode `save_depth` and the checkpoint;
  * `eval/depth_<frame>` is written at the FIRST sighting of a frame in the sweep (:343-360), i.e. normalised with
    the statistics of that batch;
  * `loss*.json["mean"]` is the mean over pairs of the per-pair losses, computed on a float32 tensor (:367-371);
  * a NaN loss skips backward/step AND the `total_iters` increment (:278-280, `continue`); `total_iters` counts
    pairs (:285) and names the validation files;
  * checkpoints every `save_epoch_freq` epochs: netG.state_dict() (:302-304).
Nothing here is reachable from consistent_depth_amd/.
"""
import json
import os
from os.path import join as pjoin

import numpy as np
import torch

from . import cpu_step, hourglass_ref, oracle


def _collate(items):
    """default_collate of VideoDataset items (video_dataset.py:179-207) -> numpy batch."""
    images = np.stack([np.asarray(it[0]) for it in items])
    g = [it[1]["geometry_consistency"] for it in items]
    batch = {
        "intrinsics": np.stack([np.asarray(it[1]["intrinsics"]) for it in items]),
        "extrinsics": np.stack([np.asarray(it[1]["extrinsics"]) for it in items]),
        "flows": [np.stack([np.asarray(x["flows"][k]) for x in g]) for k in range(2)],
        "masks": [np.stack([np.asarray(x["masks"][k]) for x in g]) for k in range(2)],
        "indices": [[int(v) for v in np.asarray(x["indices"]).tolist()] for x in g],
    }
    return images, batch


def _raw_write(fn, a):
    """utils/image_io.py:129-169 for a single-channel image: 20-byte header + fp32 rows."""
    import struct
    a = np.ascontiguousarray(a, np.float32)
    with open(fn, "wb") as f:
        f.write(struct.pack("<iiiQ", a.shape[0], a.shape[1], 5, 4))
        a.tofile(f)


class CpuLoop:
    def __init__(self, dataset, state_dict, out_dir, batch_size=4, lr=4e-4, lambda_r=1.0, lambda_b=0.1,
                 dtype=torch.float64, val_epoch_freq=1, save_epoch_freq=1, device="cpu"):
        self.ds, self.out_dir, self.bs, self.dtype = dataset, out_dir, batch_size, dtype
        self.ft = cpu_step.CpuFineTuner(state_dict, lr=lr, lambda_r=lambda_r, lambda_b=lambda_b, dtype=dtype, device=device)
        self.val_epoch_freq, self.save_epoch_freq = val_epoch_freq, save_epoch_freq
        self.total_iters = 0
        self.step_losses = []       # (epoch, pairs, loss) per training step, NaN steps included
        self.val_means = {}         # suffix -> {"reprojection": m, "disparity": m}
        os.makedirs(pjoin(out_dir, "eval"), exist_ok=True)
        os.makedirs(pjoin(out_dir, "checkpoints"), exist_ok=True)

    # ---- eval_and_save (:312-406)
    def validate(self, epoch, niters):
        suf = "_e{:04d}_iter{:06d}".format(epoch, niters)
        np_dtype = np.float64 if self.dtype == torch.float64 else np.float32
        loss_dict, saved = {"reprojection": {}, "disparity": {}}, set()
        for s0 in range(0, len(self.ds), self.bs):
            images, b = _collate([self.ds[i] for i in range(s0, min(s0 + self.bs, len(self.ds)))])
            x = torch.as_tensor(images, dtype=self.dtype).reshape((-1,) + images.shape[-3:]).to(self.ft.device)
            with torch.no_grad():      # train-mode BN, running statistics updated (see module docstring)
                pred, _ = hourglass_ref.forward(self.ft.state, x, training=True, update_running_stats=True)
            depth = torch.exp(pred).reshape(images.shape[0], 2, *pred.shape[-2:]).cpu().numpy()
            out = oracle.consistency_loss(depth, b["flows"], b["masks"], b["intrinsics"], b["extrinsics"],
                                          self.ft.lambda_r, self.ft.lambda_b, dtype=np_dtype, want_grad=False)
            for n, pair in enumerate(b["indices"]):
                for name in ("reprojection", "disparity"):
                    loss_dict[name][str(pair)] = float(out[name][n])
                for k, frame in enumerate(pair):
                    if frame in saved:
                        continue
                    saved.add(frame)
                    _raw_write(pjoin(self.out_dir, "eval", "depth_{:06d}{}.raw".format(frame, suf)), 1.0 / depth[n, k])
        means = {k: float(torch.tensor(tuple(v.values())).mean().item()) for k, v in loss_dict.items()}
        loss_dict["mean"] = means
        with open(pjoin(self.out_dir, "eval", "loss{}.json".format(suf)), "w") as f:
            json.dump(loss_dict, f)
        self.val_means[suf] = means
        return loss_dict

    # ---- the training loop (:257-310); `order(epoch)` -> list of batches, each a list of dataset indices
    def fine_tune(self, num_epochs, order, start_epoch=0):
        """`start_epoch` > 0 continues a run: no initial validation, epochs numbered from there, `order(e)` still gets
        e = 0 .. num_epochs-1 (set `total_iters` and the optimiser state first)."""
        if start_epoch == 0:
            self.validate(0, 0)
        for epoch in range(start_epoch, start_epoch + num_epochs):
            for ids in order(epoch - start_epoch):
                images, b = _collate([self.ds[i] for i in ids])
                out, _ = self.ft.step(images, b)     # NaN: no backward, no Adam step (cpu_step.py)
                loss = float(out["total"][0])
                self.step_losses.append((epoch, b["indices"], loss))
                if loss != loss:
                    continue
                self.total_iters += len(ids)
            if (epoch + 1) % self.val_epoch_freq == 0:
                self.validate(epoch + 1, self.total_iters)
            if (epoch + 1) % self.save_epoch_freq == 0:
                torch.save({k: v.detach().cpu().clone() for k, v in self.ft.state.items()},
                           pjoin(self.out_dir, "checkpoints", f"{epoch + 1:04d}.pth"))
        if (start_epoch + num_epochs) % self.val_epoch_freq != 0:
            self.validate(start_epoch + num_epochs, self.total_iters)

    # ---- save_depth (:164-199): eval-mode BN (running statistics), one frame per forward
    def save_depth(self, out_dir, frames, load_color):
        os.makedirs(pjoin(out_dir, "depth"), exist_ok=True)
        for f in frames:
            x = torch.as_tensor(np.asarray(load_color(f)), dtype=self.dtype)[None].to(self.ft.device)
            with torch.no_grad():
                pred, _ = hourglass_ref.forward(self.ft.state, x, training=False)
            _raw_write(pjoin(out_dir, "depth", "frame_{:06d}.raw".format(f)), 1.0 / torch.exp(pred)[0, 0].cpu().numpy())
