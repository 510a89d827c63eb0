"""Entry into the fine-tuning stage on precomputed inputs.

The reference's DatasetProcessor.pipeline (/root/reference/process.py:38-99) runs ten stages;
everything before "Fine-tuning" (:86) is offline CPU/third-party work whose outputs are inputs
here.  This class keeps `create_output_path` (:22-29) so directories line up, checks that the
precomputed inputs exist, then runs the two hot-path stages: fine_tune (:88) and save_depth (:93).
"""
from __future__ import annotations

import os
from os.path import join as pjoin

from . import parallel
from .depth_fine_tuning import DepthFineTuner
from .loaders.video_dataset import read_pair_list


class DatasetProcessor:
    def __init__(self, writer=None):
        self.writer = writer

    def create_output_path(self, params):
        name = f"R{params.frame_range.name}_{'-'.join(params.flow_ops)}_{params.model_type}"
        out_dir = pjoin(self.path, name)
        os.makedirs(out_dir, exist_ok=True)
        return out_dir

    def process(self, params):
        self.path = params.path
        if params.op != "all":
            raise RuntimeError(f"operation '{params.op}' is an offline stage outside this engine")
        self.out_dir = self.create_output_path(params)
        missing = [p for p in (pjoin(self.path, "color_down"), pjoin(self.path, "flow"), pjoin(self.path, "mask"),
                               pjoin(self.out_dir, "metadata_scaled.npz")) if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError("precomputed inputs missing (run the reference's offline stages or "
                                    f"tools/make_synthetic_dataset.py): {missing}")
        pairs = read_pair_list(self.path)
        frames = sorted({f for p in pairs for f in p})
        if params.frame_range.set.set is not None:
            frames = [f for f in frames if f in params.frame_range.set.set]
        print(f"Output directory: {self.out_dir}")
        ft = DepthFineTuner(self.out_dir, frames, params)
        ft.fine_tune(writer=self.writer)
        # one writer: rank 0 holds the checkpointed weights AND running statistics the exported depth must come from
        # (BatchNorm running statistics are rank-local during training, like nn.DataParallel's replica 0)
        if ft.rank == 0:
            ft.save_depth(ft.out_dir, frames)
        parallel.barrier()
        return None, ft.out_dir, frames
