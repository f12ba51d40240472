"""ORACLE package — test infrastructure only.

CPU restatements of the reference algorithms for the hot path, used as the checker by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  Product code under
``mast3r-slam-quality-dualtsdf_amd/`` must never import this package.

The C sources (``*_ref.c``) are compiled into ``oracle/liboracle.so`` by ``oracle/Makefile``
(``__graft_entry__.build()`` runs it).  This module only marshals numpy arrays into them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so if missing or stale; returns its path."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_ref.c")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


# ----------------------------------------------------------------------------------------------
# matching (oracle/matching_ref.c)
# ----------------------------------------------------------------------------------------------
def iter_proj(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh):
    """matching_kernels.cu:119-316 restated.  Returns (p_new f32[b,n,2], converged bool[b,n])."""
    rays = _c(rays_img_with_grad, np.float32)
    pts = _c(pts_3d_norm, np.float32)
    p0 = _c(p_init, np.float32)
    b, h, w, c = rays.shape
    assert c == 9
    n = pts.shape[1]
    p_new = np.zeros((b, n, 2), np.float32)
    conv = np.zeros((b, n), np.uint8)
    lib().oracle_iter_proj(
        _p(rays), _p(pts), _p(p0), _p(p_new), _p(conv),
        ctypes.c_int(b), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(n),
        ctypes.c_int(max_iter), ctypes.c_float(lambda_init), ctypes.c_float(cost_thresh),
    )
    return p_new, conv.astype(bool)


def refine_matches(D11, D21, p1, radius, dilation_max, fused_fma=False):
    """matching_kernels.cu:25-116 restated.  D11 f16[b,h,w,f], D21 f16[b,n,f], p1 i64[b,n,2]."""
    d11 = _c(D11, np.float16)
    d21 = _c(D21, np.float16)
    p = _c(p1, np.int64)
    b, h, w, f = d11.shape
    n = d21.shape[1]
    out = np.zeros((b, n, 2), np.int64)
    lib().oracle_refine_matches(
        _p(d11), _p(d21), _p(p), _p(out),
        ctypes.c_int(b), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(n), ctypes.c_int(f),
        ctypes.c_int(radius), ctypes.c_int(dilation_max), ctypes.c_int(int(fused_fma)),
    )
    return out
