"""Aggregated command line (same flags, defaults and post-parse resolution as
/root/reference/params.py:15-123).  Flags of the offline stages (video extraction, FlowNet2,
COLMAP, scale calibration, result videos) are accepted and carried in the namespace so existing
command lines keep working, but those stages are outside the accelerated hot path
(SURVEY.md section 2): the engine is entered on their precomputed outputs."""
from __future__ import annotations

import argparse

from .depth_fine_tuning import DepthFineTuningParams
from .monodepth.depth_model_registry import get_depth_model, get_depth_model_list
from .utils import frame_range, frame_sampling

# (flag, kwargs) of the pass-through groups: tools/colmap_processor.py:30-56,
# scale_calibration.py:28-34, tools/make_video.py:51-54
_PASSTHROUGH = (
    ("--colmap_bin_path", dict(default="colmap")),
    ("--sparse", dict(action="store_true")),
    ("--initialize_pose", dict(action="store_true")),
    ("--camera_params", dict(default=None)),
    ("--camera_model", dict(default="SIMPLE_PINHOLE")),
    ("--refine_intrinsics", dict(action="store_true")),
    ("--matcher", dict(choices=["exhaustive", "sequential"], default="exhaustive")),
    ("--dense_frame_ratio", dict(type=float, default=0.95)),
    ("--dense_pixel_ratio", dict(type=float, default=0.3)),
    ("--ffmpeg", dict(default="ffmpeg")),
)


# the pipeline's own flags, in the reference's order of stages (params.py:24-76)
def _stage_flags():
    return (
        ("--op", dict(choices=["all", "extract_frames"], default="all")),
        ("--path", dict(type=str, help="dataset directory (inputs and outputs)")),
        ("--video_file", dict(type=str, help="input video (ignored by the fine-tuning engine)")),
        ("--configure", dict(choices=["default", "kitti"], default="default")),
        ("--size", dict(type=int, default=384, help="long image side of the depth maps")),
        ("--align", dict(type=int, default=0, help="<= 0: the depth network's requirement")),
        ("--flow_ops", dict(nargs="*", choices=frame_sampling.SamplePairsMode.names(), default=["hierarchical2"])),
        ("--flow_checkpoint", dict(choices=["FlowNet2", "FlowNet2-KITTI"], default="FlowNet2")),
        ("--overlap_ratio", dict(type=float, default=0.2)),
    ) + _PASSTHROUGH


def _tail_flags():
    return (
        ("--model_type", dict(type=str, choices=get_depth_model_list(), default="mc")),
        ("--frame_range", dict(default="", type=frame_range.parse_frame_range, help="frames to fine-tune, e.g. 0,2-10,21-40")),
        ("--make_video", dict(action="store_true")),
        ("--seed", dict(type=int, default=0)),      # extension: seed of the epoch permutations / random init
    )


# what `--configure kitti` overrides (params.py:86-93) and which unset flags fall back to the model's class attributes
_KITTI = dict(flow_checkpoint="FlowNet2-KITTI", model_type="monodepth2", overlap_ratio=0.5, matcher="sequential")
_MODEL_DEFAULTS = (("align", lambda v: v <= 0), ("learning_rate", lambda v: v <= 0), ("lambda_view_baseline", lambda v: v < 0))


class Video3dParamsParser:
    """`Video3dParamsParser().parse(argv)` -> namespace, like the reference's class of the same name."""

    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        for flag, kw in _stage_flags():
            self.parser.add_argument(flag, **kw)
        DepthFineTuningParams.add_arguments(self.parser)
        for flag, kw in _tail_flags():
            self.parser.add_argument(flag, **kw)
        self.initialized = True

    def print(self):
        shown = {k: (v.name if isinstance(v, frame_range.NamedOptionalSet) else v) for k, v in vars(self.params).items()}
        print("------------ Parameters -------------")
        for key in sorted(shown):
            print(f"{key}: {shown[key]!r}" if isinstance(vars(self.params)[key], frame_range.NamedOptionalSet) else f"{key}: {shown[key]}")
        print("-------------------------------------")

    def parse(self, args=None, namespace=None):
        if not self.initialized:
            self.initialize()
        self.params = chosen = self.parser.parse_args(args, namespace=namespace)
        if chosen.configure == "kitti":
            vars(chosen).update(_KITTI)
        model_cls = get_depth_model(chosen.model_type)
        for name, unset in _MODEL_DEFAULTS:
            if unset(getattr(chosen, name)):
                setattr(chosen, name, getattr(model_cls, name))
        self.print()
        return chosen
