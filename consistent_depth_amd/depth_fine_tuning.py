"""Test-time fine-tuning driver on the MI355X engine.

Same surface as /root/reference/depth_fine_tuning.py: `DepthFineTuningParams.add_arguments`
(:28-63, same flags/defaults), `make_tag` (:130-136), `DepthFineTuner(range_dir, frames, params)`
with `.out_dir`, `.save_depth(dir, frames)` (:164-199), `.fine_tune(writer=None)` (:201-310) and
`.eval_and_save(...)` (:312-406); same output files:

    <out_dir>/checkpoints/%04d.pth                      netG.state_dict()
    <out_dir>/eval/depth_%06d_e%04d_iter%06d.raw        inverse depth, first sighting of a frame
    <out_dir>/eval/loss_e%04d_iter%06d.json             {loss_name: {"[i, j]": v}, "mean": {...}}
    <dir>/depth/frame_%06d.raw                          inverse depth (save_depth)

What differs, by design (MI355X-first):
  * the whole frame-pair dataset lives in HBM (loaders/pair_store.py) -- no DataLoader workers;
  * one process per GPU: the pair list is sharded over ranks (parallel.shard_indices), gradients
    are summed by ONE RCCL all-reduce of the flat buffer, per-GPU batch stays params.batch_size
    (the reference multiplies batch_size by the GPU count for nn.DataParallel, :155-159 --
    same global batch);
  * no per-step host synchronisation: the NaN-skip (:278-280) runs inside the Adam kernel,
    losses are fetched only every `print_freq` steps and at validation;
  * shuffling uses a seeded numpy permutation (the reference is unseeded); `--seed` is an
    extension flag.
PNG visualisations and tensorboard images are cosmetic and out of scope (SURVEY.md section 2
row 12); scalars are written if a `writer` with add_scalar() is passed.
"""
from __future__ import annotations

import json
import os
import time
from os.path import join as pjoin
from typing import Dict

import numpy as np
import torch

from . import optimizer, parallel
from .engine import FineTuneStep, GraphedEvaluate, GraphedFineTuneStep
from .loaders.pair_store import PairStore
from .loaders.video_dataset import VideoFrameDataset
from .loss.loss_params import LossParams
from .monodepth.depth_model_registry import get_depth_model
from .utils import image_io


class DepthFineTuningParams:
    """Fine-tuning flags (names and defaults of the reference, :36-61)."""

    @staticmethod
    def add_arguments(parser):
        LossParams.add_arguments(parser)
        parser.add_argument("--optimizer", default="Adam", choices=list(optimizer.OPTIMIZER_NAMES),
                            help="optimizer to train the network")
        parser.add_argument("--val_epoch_freq", type=int, default=1, help="validation epoch frequency.")
        parser.add_argument("--learning_rate", type=float, default=0,
                            help="<= 0 selects the default of the chosen depth model")
        parser.add_argument("--batch_size", type=int, default=4, help="frame pairs per step PER GPU")
        parser.add_argument("--num_epochs", type=int, default=20)
        parser.add_argument("--log_dir", help="folder to log tensorboard summary")
        parser.add_argument("--display_freq", type=int, default=100)
        parser.add_argument("--print_freq", type=int, default=1)
        parser.add_argument("--save_epoch_freq", type=int, default=1)
        return parser


def make_tag(params) -> str:
    return (LossParams.make_str(params) + f"_LR{params.learning_rate}" + f"_BS{params.batch_size}"
            + f"_O{params.optimizer.lower()}")


def log_loss_stats(writer, name_prefix: str, loss_meta: Dict[str, torch.Tensor], n: int, log_histogram=False):
    for sub, value in loss_meta.items():
        full = f"{name_prefix}/{sub}"
        writer.add_scalar(full + "/max", value.max(), n)
        writer.add_scalar(full + "/min", value.min(), n)
        writer.add_scalar(full + "/mean", value.mean(), n)
        if log_histogram and hasattr(writer, "add_histogram"):
            writer.add_histogram(full, value, n)


class DepthFineTuner:
    def __init__(self, range_dir, frames, params, store: PairStore = None):
        self.frames = frames
        self.params = params
        self.base_dir = params.path
        self.range_dir = range_dir
        self.out_dir = pjoin(range_dir, make_tag(params))
        self.checkpoints_dir = pjoin(self.out_dir, "checkpoints")
        self.rank, self.local_rank, self.world = parallel.env_world()
        if self.rank == 0:
            os.makedirs(self.checkpoints_dir, exist_ok=True)
            print(f"Fine-tuning directory: '{self.out_dir}'")
        self.model = get_depth_model(params.model_type)()
        print(f"Using {self.world} GPUs (one process each); global batch {params.batch_size * self.world}.")
        self.store = store
        self.seed = getattr(params, "seed", 0)
        self._resume = None

    def resume_from(self, state_dict, exp_avg, exp_avg_sq, adam_steps: int, epoch: int, total_iters: int):
        """EXTENSION (the reference cannot resume): the next fine_tune() continues a run at the START of `epoch` -- network weights
        and BatchNorm buffers from `state_dict` (a checkpoints/%04d.pth), the Adam moments as {parameter name: tensor} dicts, the
        number of optimiser steps taken and the `total_iters` (pairs) counter that names the validation files.  The validation
        sweep before the first epoch is not repeated (it belongs to the end of epoch `epoch - 1`)."""
        from .monodepth.hourglass import load_state_dict_any_prefix
        net = self.model.netG if hasattr(self.model, "netG") else self.model.model
        load_state_dict_any_prefix(net, state_dict)
        names = [n for n, _ in net.named_parameters()]
        for what, d in (("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
            missing, extra = [n for n in names if n not in d], [n for n in d if n not in set(names)]
            if missing or extra:
                raise ValueError(f"resume_from: {what} must hold exactly the model's named parameters "
                                 f"(missing {missing[:3]}{'...' if len(missing) > 3 else ''}, unknown {extra[:3]}{'...' if len(extra) > 3 else ''})")
            sizes = {n: p.numel() for n, p in net.named_parameters()}      # (flat or parameter-shaped moments alike)
            bad = [n for n in names if d[n].numel() != sizes[n]]
            if bad:
                raise ValueError(f"resume_from: {what}[{bad[0]!r}] has {d[bad[0]].numel()} elements, the parameter {sizes[bad[0]]}")
        if min(int(adam_steps), int(epoch), int(total_iters)) < 0:
            raise ValueError("resume_from: adam_steps, epoch and total_iters are counts (>= 0)")
        # the three counters of one run are tied together when no step was skipped (NaN guard): epoch * steps-per-epoch optimiser steps and
        # epoch * len(dataset) pairs.  A mismatch is legal (skipped steps lower both) but never an EXCESS -- that renames the validation files.
        n_pairs = len(self.store) if self.store is not None else None
        if n_pairs is not None and self.world == 1:
            per_epoch = -(-n_pairs // self.params.batch_size)
            if int(total_iters) > int(epoch) * n_pairs or int(adam_steps) > int(epoch) * per_epoch:
                raise ValueError(f"resume_from: epoch {epoch} of {n_pairs} pairs (batch {self.params.batch_size}) allows at most "
                                 f"total_iters {int(epoch) * n_pairs} and adam_steps {int(epoch) * per_epoch}, got {total_iters} / {adam_steps}")
        self._resume = dict(m1=[exp_avg[n] for n in names], m2=[exp_avg_sq[n] for n in names], steps=int(adam_steps),
                            epoch=int(epoch), total_iters=int(total_iters))

    # ------------------------------------------------------------------ depth export (:164-199)
    @torch.no_grad()
    def save_depth(self, dir: str = None, frames=None):
        dir = dir or self.out_dir
        frames = frames if frames is not None else self.frames
        depth_dir = pjoin(dir, "depth")
        os.makedirs(depth_dir, exist_ok=True)
        self.model.eval()
        if self.store is not None:
            rows = {f: r for r, f in enumerate(self.store.frame_ids)}
            get = lambda f: self.store.color[rows[f]][None]  # noqa: E731
        else:
            ds = VideoFrameDataset(pjoin(self.base_dir, "color_down", "frame_{:06d}.raw"), frames)
            lookup = {f: i for i, f in enumerate(frames)}
            get = lambda f: ds[lookup[f]][0][None]  # noqa: E731
        # batched eval-mode forward; inverse depth stays on the device, file writes run on a background thread
        bs = max(1, 2 * self.params.batch_size)
        with image_io.AsyncRawWriter(device=self.store.device if self.store is not None else None) as writer:
            for s0 in range(0, len(frames), bs):
                chunk = frames[s0:s0 + bs]
                images = torch.cat([get(f) for f in chunk], 0).to("cuda", non_blocking=True)
                inv = self.model.forward(images).detach().float().reciprocal()
                for i, f in enumerate(chunk):
                    writer.submit(pjoin(depth_dir, f"frame_{f:06d}.raw"), inv[i].reshape(inv.shape[-2:]))

    # ------------------------------------------------------------------ training (:201-310)
    def fine_tune(self, writer=None):
        p = self.params
        if self.store is None:
            self.store = PairStore.from_directory(self.base_dir, pjoin(self.range_dir, "metadata_scaled.npz"))
        store = self.store
        if self.world > 1:   # identical replicas by construction, not by seed: rank 0's parameters and buffers everywhere
            parallel.broadcast_([q.data for q in self.model.parameters()] + [b for b in self.model.buffers() if b.is_floating_point()])
        if getattr(self, "_step", None) is not None:
            self._step.close()               # a second fine_tune() on this object: drop the first step's parameter hooks
        step = FineTuneStep(self.model, p, world=self.world)
        if os.environ.get("CD_AMD_STEP_GRAPH", "1") != "0":   # replay the step from a HIP graph after 2 eager steps
            step = GraphedFineTuneStep(step)
        self._step = step
        if self.rank == 0:
            os.makedirs(pjoin(self.out_dir, "eval"), exist_ok=True)
        self.model.train()

        def validate(epoch, niters):
            loss_meta = self.eval_and_save(step, f"_e{epoch:04d}_iter{niters:06d}")
            if writer is not None and self.rank == 0:
                log_loss_stats(writer, "validation", loss_meta, epoch, log_histogram=True)
            if self.rank == 0:
                print(f"Done Validation for epoch {epoch} ({niters} iterations)")

        first_epoch, total_iters = 0, 0
        if self._resume is not None:
            r, self._resume = self._resume, None
            base = getattr(step, "step", step)
            by_id = {id(q): i for i, q in enumerate(self.model.parameters())}
            order = [by_id[id(q)] for q in base.opt._params]      # FlatAdam's parameter order (= model.parameters() order)
            base.opt.load_moments([r["m1"][i] for i in order], [r["m2"][i] for i in order], r["steps"])
            first_epoch, total_iters = r["epoch"], r["total_iters"]
        else:
            validate(0, 0)
        for epoch in range(first_epoch, p.num_epochs):
            t0 = time.perf_counter()
            plan = self.epoch_plan(epoch)
            plan_dev = parallel.plan_to_device(plan, store.device)   # the epoch's index lists: uploaded once
            # per-step losses of the epoch stay on the device (one 4-byte copy per step, no sync): read ONCE at the end of
            # the epoch to learn which steps the device-side NaN guard skipped
            epoch_losses = torch.zeros(max(1, len(plan)), dtype=torch.float32, device=store.device)
            sizes = self._global_step_sizes(epoch, len(plan))
            # The reference advances total_iters BEFORE it logs a step (:285-288), and its `continue` on a NaN loss skips that advance
            # (:278-280): a training point's global step = the pairs of all non-NaN steps up to and including it.  Which steps were NaN
            # is known on the host only for the print steps (whose loss is read) -- the others are read ONCE at the end of the epoch --
            # so the scalars of the print steps are buffered and written after the epoch's NaN mask is known, at their exact positions.
            pending = []        # (step index, loss value, loss_meta) of the print steps
            for it, ids in enumerate(plan):
                loss, loss_meta, metadata = step.step_from_store(store, plan_dev[it])
                epoch_losses[it:it + 1].copy_(loss.reshape(1))
                if p.print_freq > 0 and (it % max(1, p.print_freq) == 0) and self.rank == 0:
                    pairs = metadata["geometry_consistency"]["indices"].tolist()
                    lv = loss.item()  # the only host sync, every print_freq steps
                    print(f"Epoch = {epoch}, pairs = {pairs}, loss = {lv}")
                    if lv != lv:
                        print("Loss is NaN. Skipping.")  # already skipped on the device
                    elif writer is not None:
                        pending.append((it, lv, {k: v.detach().clone() for k, v in loss_meta.items()}))
            torch.cuda.synchronize()
            # the reference's `continue` on a NaN loss also skips `total_iters += batch` (:278-285); total_iters counts pairs
            # over all ranks and names the validation files.  (With world > 1 the guard acts on the all-reduced loss: a NaN on
            # any rank skips the step everywhere, so every rank subtracts the same steps after this one tiny all-reduce.)
            bad = torch.isnan(epoch_losses[:len(plan)]).float()
            if self.world > 1 and len(plan):
                parallel.allreduce_sum_(bad)
            bad = bad.cpu().numpy() > 0
            if pending:         # exact global steps of the buffered training points
                pos, at = total_iters, {}
                for i, (n, b) in enumerate(zip(sizes, bad)):
                    if not b:
                        pos += n
                    at[i] = pos
                for it, lv, meta in pending:
                    writer.add_scalar("Train/loss", lv, at[it])
                    log_loss_stats(writer, "Train/loss", meta, at[it])
            total_iters += int(sum(n for n, b in zip(sizes, bad) if not b))
            self.epoch_losses = epoch_losses[:len(plan)].cpu().numpy()
            if self.rank == 0:
                print(f"Epoch {epoch} took {time.perf_counter() - t0:.2f}s.")
            if (epoch + 1) % p.val_epoch_freq == 0:
                validate(epoch + 1, total_iters)
            if (epoch + 1) % p.save_epoch_freq == 0 and self.rank == 0:
                self.model.save(pjoin(self.checkpoints_dir, f"{epoch + 1:04d}.pth"))
        if p.num_epochs % p.val_epoch_freq != 0:
            validate(p.num_epochs, total_iters)
        if self.rank == 0:
            print("Finished Training")

    def epoch_plan(self, epoch: int):
        """This rank's per-step pair-id lists of `epoch` (store order ids).  Seeded shuffle shared by all ranks; a subclass or
        a test can override it to replay a recorded order."""
        return parallel.shard_indices(len(self.store), epoch, self.seed, self.rank, self.world, self.params.batch_size)

    def _global_step_sizes(self, epoch: int, n_steps: int):
        """Pairs processed per step over ALL ranks (for `total_iters`)."""
        if self.world == 1:
            return [len(ids) for ids in self.epoch_plan(epoch)][:n_steps]
        per_rank = [parallel.shard_indices(len(self.store), epoch, self.seed, r, self.world, self.params.batch_size) for r in range(self.world)]
        return [sum(len(pr[i]) for pr in per_rank) for i in range(n_steps)]

    # ------------------------------------------------------------------ validation (:312-406)
    @torch.no_grad()
    def eval_and_save(self, step: FineTuneStep, suf: str) -> Dict[str, torch.Tensor]:
        """Full-dataset forward + loss in store order, BatchNorm left in train mode like the
        reference (model.train() is set once, :241).  Ranks shard the unshuffled list; per-pair
        losses are gathered on rank 0, which writes the files."""
        store, p = self.store, self.params
        chunks = parallel.eval_chunks(len(store), self.rank, self.world, p.batch_size)   # whole sequential batches of the reference's sweep
        plan = [ids for _, ids in chunks]
        # the per-pair entries of ConsistencyLoss (consistency_loss.py:206); JointLoss adds that term only when one of its two
        # weights is positive (joint_loss.py:20-24), so a parameter-only configuration has no per-pair entries.  Decided from the
        # parameters, i.e. identically on every rank (the gather below needs equal column counts).
        names = ["reprojection", "disparity"] if max(p.lambda_view_baseline, p.lambda_reprojection) > 0 else []
        rows = []
        plan_dev = parallel.plan_to_device(plan, store.device)
        frames_of = store.pair_indices()    # host copy of the pair list: no device sync to learn which frames a batch holds
        first = parallel.first_sightings(frames_of, p.batch_size)   # frame -> the batch of the sweep that exports it (one owner)
        # the forward + loss of a full batch replays from a HIP graph (engine.GraphedEvaluate); its outputs are static buffers,
        # consumed below before the next batch is launched
        base = getattr(step, "step", step)
        evaluator = getattr(base, "_evaluator", None)
        if evaluator is None:
            evaluator = base._evaluator = GraphedEvaluate(base)
        with image_io.AsyncRawWriter(device=store.device) as writer:     # the files of this sweep are on disk when the block ends
            for (chunk, ids), ids_dev in zip(chunks, plan_dev):
                raw, parts, metadata = evaluator(store, ids_dev)
                idx = metadata["geometry_consistency"]["indices"]
                rows.append(torch.cat([idx.float()] + [parts[n].reshape(-1, 1).float() for n in names], 1))
                inv = self._depth_from_raw(raw).reciprocal()
                done = set()
                for b, pid in enumerate(ids):
                    for k, f in enumerate(frames_of[pid]):
                        if first[f] == chunk and f not in done:   # the reference keeps the first sighting (:343-360)
                            done.add(f)
                            writer.submit(pjoin(self.out_dir, "eval", f"depth_{f:06d}{suf}.raw"), inv[b, k])
        table = torch.cat(rows, 0) if rows else torch.zeros(0, 2 + len(names), device=store.device)
        if self.world > 1:
            import torch.distributed as dist
            n_loc = torch.tensor([table.shape[0]], device=table.device)
            n_all = [torch.zeros_like(n_loc) for _ in range(self.world)]
            dist.all_gather(n_all, n_loc)
            mx = int(max(n.item() for n in n_all))
            pad = torch.zeros(mx, table.shape[1], device=table.device)
            pad[:table.shape[0]] = table
            gathered = [torch.zeros_like(pad) for _ in range(self.world)]
            dist.all_gather(gathered, pad)
            # back into sweep order: rank r holds batches r, r + world, ...
            per_rank = [g[:int(n.item())] for g, n in zip(gathered, n_all)]
            order = []
            for r in range(self.world):
                pos = 0
                for c, ids in parallel.eval_chunks(len(store), r, self.world, p.batch_size):
                    order.append((c, per_rank[r][pos:pos + len(ids)]))
                    pos += len(ids)
            table = torch.cat([t for _, t in sorted(order, key=lambda ct: ct[0])], 0) if order else table
        table = table.cpu().numpy()
        loss_dict = {n: {} for n in (names or [])}
        for row in table:
            key = str([int(row[0]), int(row[1])])
            for c, n in enumerate(names):
                loss_dict[n][key] = float(row[2 + c])
        loss_meta = {n: torch.tensor(list(v.values())) for n, v in loss_dict.items()}
        loss_dict["mean"] = {n: float(v.mean().item()) for n, v in loss_meta.items()}
        if self.rank == 0:
            with open(pjoin(self.out_dir, "eval", f"loss{suf}.json"), "w") as f:
                json.dump(loss_dict, f)
            print("Mean: " + ", ".join(f"{n}: {v:.6f}" for n, v in loss_dict["mean"].items()))
        return loss_meta

    def _depth_from_raw(self, raw):
        from .loss.consistency_loss import DEPTH_EXP, DEPTH_RECIPROCAL
        if self.model.depth_mode == DEPTH_EXP:
            return torch.exp(raw)
        if self.model.depth_mode == DEPTH_RECIPROCAL:
            return raw.reciprocal()
        return raw
