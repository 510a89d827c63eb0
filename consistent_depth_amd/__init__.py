"""consistent_depth_amd -- MI355X-native (gfx950) engine for the test-time depth
fine-tuning hot path of facebookresearch/consistent_depth.

The compute path lives in consistent_depth_amd/csrc (hand-written HIP, C ABI declared in
include/consistent_depth_amd.h); this package is the Python host side that mirrors the
reference's plugin surface for that path:

    consistent_depth_amd.loss.{consistency_loss,joint_loss,parameter_loss,loss_params}
    consistent_depth_amd.monodepth.{depth_model,depth_model_registry,...}
    consistent_depth_amd.loaders.video_dataset, consistent_depth_amd.optimizer
    consistent_depth_amd.depth_fine_tuning, consistent_depth_amd.params, main.py

There is no CPU fallback: ops raise if the HIP library or the GPU is missing.
"""
__version__ = "0.1.0"
