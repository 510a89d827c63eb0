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


class Video3dParamsParser:
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False

    def initialize(self):
        ap = self.parser
        ap.add_argument("--op", choices=["all", "extract_frames"], default="all")
        ap.add_argument("--path", type=str, help="dataset directory (inputs and outputs)")
        ap.add_argument("--video_file", type=str, help="input video (ignored by the fine-tuning engine)")
        ap.add_argument("--configure", choices=["default", "kitti"], default="default")
        # video
        ap.add_argument("--size", type=int, default=384, help="long image side of the depth maps")
        ap.add_argument("--align", type=int, default=0, help="<= 0: the depth network's requirement")
        # flow
        ap.add_argument("--flow_ops", nargs="*", choices=frame_sampling.SamplePairsMode.names(),
                        default=["hierarchical2"])
        ap.add_argument("--flow_checkpoint", choices=["FlowNet2", "FlowNet2-KITTI"], default="FlowNet2")
        ap.add_argument("--overlap_ratio", type=float, default=0.2)
        # calibration / make-video groups (accepted, unused by the engine)
        for flag, kw in _PASSTHROUGH:
            ap.add_argument(flag, **kw)
        # fine-tuning
        DepthFineTuningParams.add_arguments(ap)
        ap.add_argument("--model_type", type=str, choices=get_depth_model_list(), default="mc")
        ap.add_argument("--frame_range", default="", type=frame_range.parse_frame_range,
                        help="frames to fine-tune, e.g. 0,2-10,21-40")
        ap.add_argument("--make_video", action="store_true")
        # extension (not in the reference): seed of the epoch permutations / random init
        ap.add_argument("--seed", type=int, default=0)
        self.initialized = True

    def print(self):
        print("------------ Parameters -------------")
        for k, v in sorted(vars(self.params).items()):
            print(f"{k}: '{v.name}'" if isinstance(v, frame_range.NamedOptionalSet) else f"{k}: {v}")
        print("-------------------------------------")

    def parse(self, args=None, namespace=None):
        if not self.initialized:
            self.initialize()
        self.params = p = self.parser.parse_args(args, namespace=namespace)
        if p.configure == "kitti":
            p.flow_checkpoint, p.model_type, p.overlap_ratio, p.matcher = "FlowNet2-KITTI", "monodepth2", 0.5, "sequential"
        model = get_depth_model(p.model_type)
        if p.align <= 0:
            p.align = model.align
        if p.learning_rate <= 0:
            p.learning_rate = model.learning_rate
        if p.lambda_view_baseline < 0:
            p.lambda_view_baseline = model.lambda_view_baseline
        self.print()
        return p
