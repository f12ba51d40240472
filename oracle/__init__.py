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


# ----------------------------------------------------------------------------------------------
# Gauss-Newton backend (oracle/gn_ref.c)
# ----------------------------------------------------------------------------------------------
KIND = {"rays": 0, "calib": 1, "points": 2}


def _ci(x):
    return ctypes.c_int(int(x))


def _cf(x):
    return ctypes.c_float(float(x))


def edge_rows(ii, jj, num_fix=1):
    """get_unique_kf_idx + create_inds (gn_kernels.cu:161-170): (unique, ii_edge, jj_edge, ii_opt, jj_opt)."""
    ii = np.asarray(ii, np.int64)
    jj = np.asarray(jj, np.int64)
    uniq = np.unique(np.concatenate((ii, jj)))
    ie = np.searchsorted(uniq, ii).astype(np.int64)
    je = np.searchsorted(uniq, jj).astype(np.int64)
    return uniq, ie, je, ie - num_fix, je - num_fix


def gn_edges(kind, Twc, Xs, Cs, K, ii_edge, jj_edge, idx_ii2jj, valid_match, Q, sigma_a, sigma_b, C_thresh,
             Q_thresh, height=0, width=1, pixel_border=0, z_eps=0.0):
    """One launch of the reference's edge kernel: returns Hs (4,E,7,7), gs (2,E,7)."""
    Twc = _c(Twc, np.float32); Xs = _c(Xs, np.float32); Cs = _c(Cs, np.float32)
    ie = _c(ii_edge, np.int64); je = _c(jj_edge, np.int64)
    idx = _c(idx_ii2jj, np.int64); vm = _c(valid_match, np.uint8); Qc = _c(Q, np.float32)
    E, HW = idx.shape[0], Xs.shape[1]
    Hs = np.zeros((4, E, 7, 7), np.float32)
    gs = np.zeros((2, E, 7), np.float32)
    Kc = _c(K, np.float32) if K is not None else None
    lib().oracle_gn_edges(
        _ci(KIND[kind]), _p(Twc), _p(Xs), _p(Cs), _p(Kc) if Kc is not None else None, _p(ie), _p(je), _p(idx),
        _p(vm), _p(Qc), _ci(HW), _ci(E), _cf(sigma_a), _cf(sigma_b), _cf(C_thresh), _cf(Q_thresh), _ci(height),
        _ci(width), _ci(pixel_border), _cf(z_eps), _p(Hs), _p(gs))
    return Hs, gs


def gn_solve(Hs, gs, ii_opt, jj_opt, N):
    """SparseBlock assemble + LL^T solve (gn_kernels.cu:57-159): returns (dx (N,7) f32, failed)."""
    Hs = _c(Hs, np.float32); gs = _c(gs, np.float32)
    io = _c(ii_opt, np.int64); jo = _c(jj_opt, np.int64)
    dx = np.zeros((N, 7), np.float32)
    lib().oracle_gn_solve.restype = ctypes.c_int
    fail = lib().oracle_gn_solve(_p(Hs), _p(gs), _p(io), _p(jo), _ci(io.shape[0]), _ci(N), _p(dx))
    return dx, bool(fail)


def gauss_newton(kind, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, sigma_a, sigma_b, C_thresh, Q_thresh,
                 max_iter, delta_thresh, height=0, width=1, pixel_border=0, z_eps=0.0):
    """Full host loop (gn_kernels.cu:1140-1228 etc.).  Returns (Twc_new, dx, iterations)."""
    Twc = np.array(Twc, np.float32, copy=True, order="C")
    Xs = _c(Xs, np.float32); Cs = _c(Cs, np.float32)
    ii = _c(ii, np.int64); jj = _c(jj, np.int64)
    idx = _c(idx_ii2jj, np.int64); vm = _c(valid_match, np.uint8); Qc = _c(Q, np.float32)
    P, HW = Xs.shape[0], Xs.shape[1]
    E = ii.shape[0]
    dx = np.zeros((P - 1, 7), np.float32)
    Kc = _c(K, np.float32) if K is not None else None
    lib().oracle_gauss_newton.restype = ctypes.c_int
    iters = lib().oracle_gauss_newton(
        _ci(KIND[kind]), _p(Twc), _p(Xs), _p(Cs), _p(Kc) if Kc is not None else None, _p(ii), _p(jj), _p(idx),
        _p(vm), _p(Qc), _ci(P), _ci(HW), _ci(E), _cf(sigma_a), _cf(sigma_b), _cf(C_thresh), _cf(Q_thresh),
        _ci(height), _ci(width), _ci(pixel_border), _cf(z_eps), _ci(max_iter), _cf(delta_thresh), _p(dx))
    return Twc, dx, iters


def sim3_exp(xi):
    xi = _c(xi, np.float32).reshape(-1, 7)
    out = np.zeros((xi.shape[0], 8), np.float32)
    for k in range(xi.shape[0]):
        lib().oracle_sim3_exp(_p(xi[k]), _p(out[k]))
    return out


def sim3_retr(xi, T):
    xi = _c(xi, np.float32).reshape(-1, 7); T = _c(T, np.float32).reshape(-1, 8)
    out = np.zeros_like(T)
    for k in range(T.shape[0]):
        lib().oracle_sim3_retr(_p(xi[k]), _p(T[k]), _p(out[k]))
    return out


def sim3_retr_rows(dx, Twc, num_fix=1):
    """pose_retr_kernel (gn_kernels.cu:415-453): rows >= num_fix of Twc retracted by dx (P-num_fix,7)."""
    out = np.array(Twc, np.float32, copy=True)
    out[num_fix:] = sim3_retr(dx, out[num_fix:])
    return out


def sim3_rel(Ti, Tj):
    Ti = _c(Ti, np.float32).reshape(-1, 8); Tj = _c(Tj, np.float32).reshape(-1, 8)
    out = np.zeros_like(Ti)
    for k in range(Ti.shape[0]):
        lib().oracle_sim3_rel(_p(Ti[k]), _p(Tj[k]), _p(out[k]))
    return out


def sim3_act(T, X):
    T = _c(T, np.float32).reshape(8); X = _c(X, np.float32).reshape(-1, 3)
    Y = np.zeros_like(X)
    lib().oracle_sim3_act(_p(T), _p(X), _p(Y), _ci(X.shape[0]))
    return Y


def sim3_adj_inv(T, x7):
    T = _c(T, np.float32).reshape(8); x7 = _c(x7, np.float32).reshape(7)
    y = np.zeros(7, np.float32)
    lib().oracle_sim3_adj_inv(_p(T), _p(x7), _p(y))
    return y


# ----------------------------------------------------------------------------------------------
# global sparse TSDF (oracle/tsdf_ref.c)
# ----------------------------------------------------------------------------------------------
class TSDFVolume:
    """Restatement of mast3r_slam/tsdf/global_volume.py:15-140 (same constructor / method names)."""

    def __init__(self, voxel_size, truncation, max_weight=100.0, min_weight=1.0e-3):
        L = lib()
        L.oracle_tsdf_create.restype = ctypes.c_void_p
        L.oracle_tsdf_size.restype = ctypes.c_size_t
        self.voxel_size, self.truncation = float(voxel_size), float(truncation)
        self._h = ctypes.c_void_p(L.oracle_tsdf_create(
            ctypes.c_double(voxel_size), ctypes.c_double(truncation), ctypes.c_double(max_weight),
            ctypes.c_double(min_weight)))

    def __del__(self):
        try:
            lib().oracle_tsdf_free(self._h)
        except Exception:
            pass

    def integrate(self, points_world, confidences, cam_origin, step_scale=0.5):
        pts = _c(points_world, np.float32).reshape(-1, 3)
        conf = _c(confidences, np.float64).reshape(-1)
        org = _c(cam_origin, np.float32).reshape(3)
        if pts.size == 0:
            return 0
        lib().oracle_tsdf_integrate.restype = ctypes.c_int
        return lib().oracle_tsdf_integrate(self._h, _p(pts), _p(conf), _p(org), _ci(pts.shape[0]),
                                           ctypes.c_double(step_scale))

    def voxels(self):
        """(keys i64[n,3], tsdf f64[n], weight f64[n]) sorted lexicographically by key."""
        n = lib().oracle_tsdf_size(self._h)
        keys = np.zeros((n, 3), np.int64); t = np.zeros(n); w = np.zeros(n)
        if n:
            lib().oracle_tsdf_dump(self._h, _p(keys), _p(t), _p(w))
            o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
            keys, t, w = keys[o], t[o], w[o]
        return keys, t, w

    def query(self, points):
        """Vectorised query: (value f64[n], grad f64[n,3], status u8[n]); status 0 None/None,
        1 value only, 2 value+gradient."""
        pts = _c(points, np.float32).reshape(-1, 3)
        n = pts.shape[0]
        val = np.zeros(n); g = np.zeros((n, 3)); st = np.zeros(n, np.uint8)
        lib().oracle_tsdf_query(self._h, _p(pts), _ci(n), _p(val), _p(g), _p(st))
        return val, g, st

    def pose_system(self, points_world, conf, lam):
        pts = _c(points_world, np.float32).reshape(-1, 3)
        cf = _c(conf, np.float32).reshape(-1)
        H = np.zeros((7, 7)); b = np.zeros(7)
        lib().oracle_tsdf_pose_system.restype = ctypes.c_int
        used = lib().oracle_tsdf_pose_system(self._h, _p(pts), _p(cf), _ci(pts.shape[0]), ctypes.c_double(lam),
                                             _p(H), _p(b))
        return H, b, used
