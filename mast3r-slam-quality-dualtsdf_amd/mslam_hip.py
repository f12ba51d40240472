"""ctypes binding of libmslam_hip.so (C ABI declared in include/mslam_hip.h).

This is the only place that touches the shared library.  There is deliberately NO fallback: if the
library is missing, or a tensor is not resident on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSLAM_LIB", os.path.join(_HERE, "libmslam_hip.so"))   # MSLAM_LIB: measurement builds (tools/probes)

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_vp = ctypes.c_void_p
_c_i64 = ctypes.c_int64
_c_double = ctypes.c_double
_c_size = ctypes.c_size_t

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "mslam_abi_version": [],
    "mslam_device_check": [],
    "mslam_iter_proj": [_c_vp] * 5 + [_c_int] * 5 + [_c_float, _c_float, _c_vp],
    "mslam_refine_matches": [_c_vp] * 4 + [_c_int] * 7 + [_c_vp],
    "mslam_prep_iter_proj": [_c_vp] * 6 + [_c_int] * 3 + [_c_vp],
    "mslam_match_occlusion": [_c_vp] * 5 + [_c_int] * 3 + [_c_float, _c_vp],
    "mslam_pixel_to_lin": [_c_vp] * 2 + [_c_int] * 3 + [_c_vp],
    "mslam_gauss_newton_rays": [_c_vp] * 8 + [_c_int] * 3 + [_c_float] * 4 + [_c_int, _c_float, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_gauss_newton_calib": [_c_vp] * 9 + [_c_int] * 6 + [_c_float] * 5 + [_c_int, _c_float, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_gauss_newton_points": [_c_vp] * 8 + [_c_int] * 3 + [_c_float] * 3 + [_c_int, _c_float, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_gn_begin": [_c_vp] * 2 + [_c_int] * 3 + [_c_vp, _c_size, _c_vp],
    "mslam_gn_compact": [_c_vp] * 5 + [_c_int] * 5 + [_c_float] * 2 + [_c_vp, _c_size, _c_vp],
    "mslam_gn_compact_at": [_c_vp] * 5 + [_c_int] * 7 + [_c_float] * 2 + [_c_vp, _c_size, _c_vp],
    "mslam_gn_accumulate": [_c_int] + [_c_vp] * 2 + [_c_int] * 5 + [_c_float] * 2 + [_c_int] * 3 + [_c_float] + [_c_vp] * 3 + [_c_size, _c_vp],
    "mslam_gn_solve_retract": [_c_vp] * 2 + [_c_int] * 3 + [_c_vp, _c_vp, _c_float, _c_vp, _c_size, _c_vp],
    "mslam_gn_status": [_c_vp] + [_c_int] * 3 + [_c_vp, _c_size, _c_vp],
    "mslam_sim3_act": [_c_vp] * 3 + [_c_int, ctypes.c_longlong, _c_int, _c_vp],
    "mslam_sim3_op": [_c_int] + [_c_vp] * 3 + [_c_int] * 3 + [_c_vp],
    "mslam_mast3r_create": [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp],
    "mslam_mast3r_destroy": [_c_vp],
    "mslam_mast3r_encode": [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_mast3r_decode": [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 10 + [_c_vp, _c_size, _c_vp],
    "mslam_gemm_bf16": [_c_vp] * 5 + [_c_int] * 5 + [_c_vp],
    "mslam_gemm_tile_override": [_c_int] * 4,
    "mslam_gemm_profile_begin": [_c_int] * 4,
    "mslam_gemm_profile_end": [_c_vp] * 3,
    "mslam_conv2d_nhwc_bf16": [_c_vp] * 5 + [_c_int] * 9 + [_c_vp],
    "mslam_attention_bf16": [_c_vp] * 4 + [_c_int] * 4 + [_c_vp],
    "mslam_layernorm_f32": [_c_vp] * 5 + [_c_int, _c_int, _c_float, _c_vp],
    "mslam_track_pose": [_c_int] + [_c_vp] * 6 + [_c_int, _c_vp, _c_int, _c_int] + [_c_float] * 3 + [_c_int, _c_float, _c_int, _c_int, _c_float, _c_float, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_track_prepare": [_c_vp] * 5 + [_c_float, _c_vp, _c_float, _c_float, _c_float, _c_int] + [_c_vp] * 8 + [_c_size, _c_vp],
    "mslam_track_verdict": [_c_vp, _c_vp, _c_int, _c_vp, _c_vp],
    "mslam_track_fuse": [_c_vp] * 6 + [_c_int] + [_c_vp] * 4,
    "mslam_tsdf_local_build": [_c_vp] * 5 + [_c_int] * 4 + [_c_double, _c_double, _c_float, _c_vp, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_tsdf_local_raycast": [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_float, _c_vp, _c_vp, _c_vp],
    "mslam_quality_reduce_grid": [_c_vp] * 3 + [_c_int] * 4 + [_c_double, _c_double, _c_vp, _c_vp],
    "mslam_quality_classify": [_c_vp] * 3 + [_c_int] + [_c_float] * 3 + [_c_vp] * 3,
    "mslam_tsdf_table_init": [_c_vp, _c_size, ctypes.c_uint64, _c_vp],
    "mslam_tsdf_integrate": [_c_vp, ctypes.c_uint64, _c_vp, _c_vp, _c_vp, _c_int] + [_c_double] * 4 + [_c_int, _c_int, _c_vp, _c_size, _c_vp],
    "mslam_room_pair": [_c_vp, _c_vp] + [_c_int] * 4 + [_c_double] * 5 + [_c_vp] * 10 + [_c_vp],
    "mslam_remap_bilinear_u8": [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp],
    "mslam_gemm_f64": [_c_vp, _c_int, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp],
    "mslam_asmk_aggregate": [_c_vp] * 5 + [_c_int] * 5 + [_c_vp],
    "mslam_asmk_search": [_c_vp] * 3 + [_c_int] + [_c_vp] * 2 + [_c_int] * 2 + [_c_float] * 2 + [_c_vp, _c_vp],
    "mslam_tsdf_rehash": [_c_vp, ctypes.c_uint64, _c_vp, ctypes.c_uint64, _c_vp],
    "mslam_tsdf_header": [_c_vp, ctypes.c_uint64, _c_vp, _c_vp],
    "mslam_tsdf_dump": [_c_vp, ctypes.c_uint64, _c_vp, _c_vp, _c_vp, ctypes.c_uint32, _c_vp],
    "mslam_tsdf_query": [_c_vp, ctypes.c_uint64, _c_vp, _c_int, _c_double, _c_double, _c_vp, _c_vp, _c_vp, _c_vp],
    "mslam_tsdf_lookup7": [_c_vp, ctypes.c_uint64, _c_vp, _c_int, _c_vp, _c_double, _c_vp, _c_vp],
    "mslam_tsdf_query_lookup": [_c_vp, _c_int, _c_double, _c_double, _c_vp, _c_vp, _c_vp, _c_vp],
    "mslam_tsdf_pose_step_lookup": [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_double] * 4 + [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_size, _c_vp],
    "mslam_tsdf_pose_step": [_c_vp, ctypes.c_uint64, _c_vp, _c_vp, _c_int, _c_vp, _c_int] + [_c_double] * 4 + [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_size, _c_vp],
}
_RESTYPES = {
    "mslam_last_error": ctypes.c_char_p,
    "mslam_gn_workspace_bytes": ctypes.c_size_t,
    "mslam_tsdf_table_bytes": ctypes.c_size_t,
    "mslam_tsdf_local_workspace_bytes": ctypes.c_size_t,
    "mslam_track_workspace_bytes": ctypes.c_size_t,
    "mslam_track_prepare_workspace_bytes": ctypes.c_size_t,
    "mslam_mast3r_workspace_bytes": ctypes.c_size_t,
    "mslam_tsdf_integrate_workspace_bytes": ctypes.c_size_t,
}

_lib = None


def exported_symbols():
    """Every symbol include/mslam_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGNATURES) + list(_RESTYPES))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        handle.mslam_last_error.argtypes = []
        handle.mslam_last_error.restype = ctypes.c_char_p
        handle.mslam_gn_workspace_bytes.argtypes = [_c_int] * 4
        handle.mslam_gn_workspace_bytes.restype = ctypes.c_size_t
        handle.mslam_mast3r_workspace_bytes.argtypes = [_c_vp, _c_int, _c_int, _c_int]
        handle.mslam_mast3r_workspace_bytes.restype = ctypes.c_size_t
        handle.mslam_track_workspace_bytes.argtypes = [_c_int]
        handle.mslam_track_workspace_bytes.restype = ctypes.c_size_t
        handle.mslam_track_prepare_workspace_bytes.argtypes = [_c_int]
        handle.mslam_track_prepare_workspace_bytes.restype = ctypes.c_size_t
        handle.mslam_tsdf_local_workspace_bytes.argtypes = [_c_int]
        handle.mslam_tsdf_local_workspace_bytes.restype = ctypes.c_size_t
        handle.mslam_tsdf_table_bytes.argtypes = [ctypes.c_uint64]
        handle.mslam_tsdf_table_bytes.restype = ctypes.c_size_t
        handle.mslam_tsdf_integrate_workspace_bytes.argtypes = [_c_int, _c_double, _c_double, _c_double]
        handle.mslam_tsdf_integrate_workspace_bytes.restype = ctypes.c_size_t
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mslam_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL).  Raises for host tensors: no CPU path exists."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError(
            "libmslam_hip.so operates on HIP device tensors only; got a tensor on "
            f"{t.device}. There is no CPU fallback."
        )
    return t.data_ptr()


def require_contiguous(**tensors) -> None:
    """Mirror of CHECK_CONTIGUOUS (mast3r_slam/backend/include/gn.h:5): RuntimeError by name."""
    for name, t in tensors.items():
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")


def require_dtype(t, dtype, name: str) -> None:
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
