"""`monodepth2` registry entry (reference: /root/reference/monodepth/monodepth2_model.py:15-93).

Out of scope for acceleration (KITTI-only adapter, its network is
an un-vendored submodule and its save() is a no-op in the reference).  The registry name and
the class attributes used by params.py are kept; constructing it says so explicitly.
"""
from .depth_model import DepthModel


class Monodepth2Model(DepthModel):
    align = 1
    learning_rate = 0.00004
    lambda_view_baseline = 1

    def __init__(self):
        super().__init__()
        raise NotImplementedError(
            "monodepth2 is outside the accelerated hot path of consistent_depth_amd (SURVEY.md section 8); "
            "use --model_type mc or midas2")

    def estimate_depth(self, images):
        raise NotImplementedError

    def save(self, file_name):
        pass
