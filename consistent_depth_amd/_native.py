"""ctypes binding of the C-ABI library (include/consistent_depth_amd.h).

The product path has NO fallback: if libcd_amd.so is missing or a tensor is not a
contiguous fp32 tensor on the HIP device, this module raises.
"""
from __future__ import annotations

import ctypes
import os

import torch  # imported first on purpose: libcd_amd.so then binds to torch's libamdhip64

_PKG = os.path.dirname(os.path.abspath(__file__))
# CD_AMD_LIB: load another build of the SAME library (A/B measurements of kernel variants, tools/exp/build_variants.sh); not a fallback
SO_PATH = os.environ.get("CD_AMD_LIB") or os.path.join(_PKG, "libcd_amd.so")
ABI_VERSION = 9
BN_STAT_SLOTS = 16   # CD_BN_STAT_SLOTS of include/consistent_depth_amd.h (checked by tests/test_abi.py)

_lib = None

c_f, c_i, c_p, c_sz = ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes): exactly the declarations of include/consistent_depth_amd.h
SIGNATURES = {
    "cd_abi_version": (c_i, []),
    "cd_build_info": (ctypes.c_char_p, []),
    "cd_consistency_loss_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "cd_consistency_loss_workspace_init": (c_i, [c_p, c_sz, c_p]),
    "cd_mask_sums": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "cd_tile_windows_bytes": (c_sz, [c_i, c_i, c_i]),
    "cd_tile_windows": (c_i, [c_p] * 4 + [c_i, c_i, c_i, c_p, c_p]),
    "cd_debug_set_overflow_capacity": (c_i, [c_i]),
    "cd_debug_set_loss_variant": (c_i, [c_i]),
    "cd_debug_set_loss_chunk": (c_i, [c_i]),
    "cd_debug_set_loss_sweep": (c_i, [c_i]),
    "cd_consistency_loss_fwd_bwd": (c_i, [c_p] * 9 + [c_f, c_f, c_i, c_i, c_i, c_i] + [c_p] * 4 + [c_p, c_sz, c_p]),
    "cd_consistency_loss_fwd": (c_i, [c_p] * 8 + [c_f, c_f, c_i, c_i, c_i, c_i] + [c_p] * 3 + [c_p, c_sz, c_p]),
    "cd_profile_begin": (c_i, [c_i]),
    "cd_profile_end": (c_i, [c_p, c_p, c_i, c_p]),
    "cd_gather_pairs": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "cd_sample_bilinear_border": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "cd_flow_consistency_masks": (c_i, [c_p, c_p, c_p, c_p, c_i, ctypes.c_double, ctypes.c_double, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cd_warp_image": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cd_depth_to_points": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "cd_frame_median_scales": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "cd_conv2d_packed_weight_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "cd_conv2d_pack_weights": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "cd_conv2d_pack_weights_table": (c_i, [c_p, c_i, c_p]),
    "cd_conv2d_fwd": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_conv2d_fwd_cfg": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_conv2d_packed_co_tiles": (c_i, [c_i, c_i]),
    "cd_conv2d_fwd_grouped": (c_i, [c_p, c_i, c_i, c_i, c_p, c_sz, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_conv2d_fwd_multi": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "cd_conv2d_wgrad_grouped": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_sz, c_i, c_i, c_i, c_i, c_p]),
    "cd_set_conv_arith": (c_i, [c_i]),
    "cd_get_conv_arith": (c_i, []),
    "cd_debug_force_conv_tile_rows": (c_i, [c_i]),
    "cd_debug_force_conv_co_tiles": (c_i, [c_i]),
    "cd_debug_set_conv_pipeline": (c_i, [c_i]),
    "cd_debug_set_wgrad_mode": (c_i, [c_i]),
    "cd_conv2d_wgrad_workspace_floats": (c_sz, [c_i, c_i, c_i]),
    "cd_conv2d_wgrad": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cd_conv2d_wgrad_desc": (c_i, [c_p]),
    "cd_conv2d_wgrad_table": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "cd_conv2d_wgrad_plan": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)]),
    "cd_conv2d_wgrad_unpack_table": (c_i, [c_p, c_i, c_p]),
    "cd_rccl_available": (c_i, []),
    "cd_allreduce_mean_f32": (c_i, [c_p, c_sz, c_p, c_i, c_p]),
    "cd_hourglass_create": (c_i, [c_i, c_i, c_i, ctypes.POINTER(c_p)]),
    "cd_hourglass_destroy": (c_i, [c_p]),
    "cd_hourglass_param_floats": (c_sz, [c_p]),
    "cd_hourglass_bn_floats": (c_sz, [c_p]),
    "cd_hourglass_param_count": (c_i, [c_p]),
    "cd_hourglass_param_info": (c_i, [c_p, c_i, ctypes.POINTER(c_sz), ctypes.POINTER(c_i)]),
    "cd_hourglass_params": (c_p, [c_p]),
    "cd_hourglass_grads": (c_p, [c_p]),
    "cd_hourglass_load_state": (c_i, [c_p, c_p, c_p, c_p]),
    "cd_hourglass_save_state": (c_i, [c_p, c_p, c_p, c_p]),
    "cd_hourglass_zero_grad": (c_i, [c_p, c_p]),
    "cd_copy_f32": (c_i, [c_p, c_p, c_sz, c_p]),
    "cd_hourglass_forward": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "cd_hourglass_backward": (c_i, [c_p, c_p, c_p]),
    "cd_bn_normalize": (c_i, [c_p, c_i, c_i, c_i, c_p, c_f, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "cd_bn_finalize": (c_i, [c_p, c_i, c_i, c_i, ctypes.c_double, c_f, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    "cd_bn_relu_bwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p]),
    "cd_avgpool2_fwd": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_avgpool2_bwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_upsample2x_add_fwd": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_upsample2x_bwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_upsample2x_halfpixel_fwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_upsample2x_halfpixel_bwd": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_bn_block_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cd_bn_block_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cd_eltwise": (c_i, [c_p, c_p, c_p, ctypes.c_size_t, c_i, c_p]),
    "cd_maxpool3s2_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cd_maxpool3s2_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "cd_add_slice": (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "cd_debug_set_layers_mode": (c_i, [c_i]),
    "cd_copy_segments": (c_i, [c_p, c_i, c_p]),
    "cd_counters_add": (c_i, [c_p, c_i, ctypes.c_longlong, c_p]),
    "cd_zero_bytes": (c_i, [c_p, c_sz, c_p]),
    "cd_channel_sum": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "cd_adam_step_flat": (c_i, [c_p] * 4 + [c_sz, c_f, c_f, c_f, c_f, c_i, c_f, c_p]),
    "cd_adam_step_flat_guarded": (c_i, [c_p] * 4 + [c_sz, c_f, c_f, c_f, c_f, c_p, c_p, c_f, c_p]),
    "cd_l1_distance_workspace_bytes": (c_sz, [c_sz]),
    "cd_l1_distance": (c_i, [c_p, c_p, c_sz, c_p, c_p, c_sz, c_p]),
}

STATUS = {0: "CD_OK", -1: "CD_ERR_INVALID_ARG", -2: "CD_ERR_WORKSPACE", -3: "CD_ERR_LAUNCH", -4: "CD_ERR_UNSUPPORTED"}


class NativeLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libcd_amd.so (once).  Raises NativeLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NativeLibraryError(
            f"{SO_PATH} is missing: build it with `python -m consistent_depth_amd.build_native` "
            "(hipcc --offload-arch=gfx950).  consistent_depth_amd has no CPU/PyTorch fallback.")
    l = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{SO_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    if l.cd_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI mismatch: library {l.cd_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = l
    return l


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {STATUS.get(rc, rc)}")


def dev_ptr(t: torch.Tensor, name: str = "tensor") -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live on the HIP device (got {t.device}); "
                           "consistent_depth_amd has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    return t.data_ptr()


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def zero_(t: torch.Tensor) -> torch.Tensor:
    """t.zero_() by a kernel of the library on the current stream (contiguous device tensors; no framework kernel inside a captured step)."""
    if not (t.is_cuda and t.is_contiguous()):
        raise RuntimeError("zero_: a contiguous tensor on the HIP device")
    if t.data_ptr() % 16 or (t.numel() * t.element_size()) % 4:
        return t.zero_()       # (a view at an odd offset: none in the step)
    if t.numel():
        check(lib().cd_zero_bytes(t.data_ptr(), t.numel() * t.element_size(), stream_ptr(t.device)), "cd_zero_bytes")
    return t


_workspaces: dict = {}
_retired: list = []   # outgrown workspaces stay alive: a captured HIP graph may have their addresses baked in


def workspace(key, nbytes: int, device, on_new=None) -> torch.Tensor:
    """Persistent per-(key, device) scratch buffer, grown on demand (never shrunk, never freed: a step graph captured
    with the smaller buffer keeps replaying into it while later, larger calls use the new one).  on_new(buf): called once
    for every buffer this function allocates (cd_consistency_loss_workspace_init for the loss workspace)."""
    k = (key, str(device))
    buf = _workspaces.get(k)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        if on_new is not None:
            on_new(buf)
        _workspaces[k] = buf
    return buf
