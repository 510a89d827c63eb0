"""Depth-model plugin registry: short name -> model class.

The reference resolves the `--model_type` flag through three functions (/root/reference/monodepth/depth_model_registry.py:
`get_depth_model_list`, `get_depth_model`, `create_depth_model`; unknown names raise ValueError with the same message).
Here the table maps each name to "module:Class" and the module is imported on first use, so selecting `mc` never imports
the MiDaS backbone (and its 105 M parameters' worth of torch modules) and vice versa."""
from __future__ import annotations

import importlib
from typing import Dict, List, Type

from .depth_model import DepthModel

_PLUGINS: Dict[str, str] = {
    "mc": ".mannequin_challenge_model:MannequinChallengeModel",
    "midas2": ".midas_v2_model:MidasV2Model",
    "monodepth2": ".monodepth2_model:Monodepth2Model",
}


def get_depth_model_list() -> List[str]:
    return list(_PLUGINS)


def get_depth_model(type: str) -> Type[DepthModel]:
    try:
        module, cls = _PLUGINS[type].split(":")
    except KeyError:
        raise ValueError(f"Unsupported model type '{type}'.") from None
    return getattr(importlib.import_module(module, package=__package__), cls)


def create_depth_model(type: str) -> DepthModel:
    return get_depth_model(type)()
