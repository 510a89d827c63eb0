"""String -> depth-model class registry (mirrors
/root/reference/monodepth/depth_model_registry.py:12-29: same names, same ValueError)."""
from typing import List

from .depth_model import DepthModel

_NAMES = ("mc", "midas2", "monodepth2")


def get_depth_model_list() -> List[str]:
    return list(_NAMES)


def get_depth_model(type: str) -> DepthModel:
    if type == "mc":
        from .mannequin_challenge_model import MannequinChallengeModel
        return MannequinChallengeModel
    if type == "midas2":
        from .midas_v2_model import MidasV2Model
        return MidasV2Model
    if type == "monodepth2":
        from .monodepth2_model import Monodepth2Model
        return Monodepth2Model
    raise ValueError(f"Unsupported model type '{type}'.")


def create_depth_model(type: str) -> DepthModel:
    return get_depth_model(type)()
