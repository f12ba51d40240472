"""Drop-in for the reference's pybind module ``mast3r_slam_backends``
(mast3r_slam/backend/src/gn.cpp:116-123), backed by libmslam_hip.so on MI355X.

Same function names, argument order/meaning, dtypes, return lists and error behaviour
(non-contiguous input -> RuntimeError("<name> must be contiguous")).  Outputs are freshly
allocated torch tensors on the inputs' device, launched on torch's current HIP stream.
"""
import torch

import mslam_hip as _m


def iter_proj(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh):
    """gn.cpp:84-99 / matching_kernels.cu:279-316.
    f32[b,h,w,9], f32[b,n,3], f32[b,n,2] -> [p_new f32[b,n,2], converged bool[b,n]]"""
    _m.require_contiguous(rays_img_with_grad=rays_img_with_grad, pts_3d_norm=pts_3d_norm, p_init=p_init)
    for name, t in (("rays_img_with_grad", rays_img_with_grad), ("pts_3d_norm", pts_3d_norm), ("p_init", p_init)):
        _m.require_dtype(t, torch.float32, name)
    b, h, w, c = rays_img_with_grad.shape
    if c != 9:
        raise RuntimeError(f"rays_img_with_grad must have 9 channels, got {c}")
    n = p_init.shape[1]
    # the reference allocates with torch::zeros (matching_kernels.cu:295-301); the kernel writes every element, so
    # the fill launches are skipped here
    p_new = torch.empty((b, n, 2), dtype=p_init.dtype, device=p_init.device)
    converged = torch.empty((b, n), dtype=torch.bool, device=p_init.device)
    rc = _m.lib().mslam_iter_proj(
        _m.ptr(rays_img_with_grad), _m.ptr(pts_3d_norm), _m.ptr(p_init), _m.ptr(p_new), _m.ptr(converged),
        b, h, w, n, int(max_iter), float(lambda_init), float(cost_thresh), _m.stream_ptr(),
    )
    _m.check(rc, "iter_proj")
    return [p_new, converged]


def refine_matches(D11, D21, p1, radius, dilation_max):
    """gn.cpp:101-114 / matching_kernels.cu:84-116.
    half[b,h,w,f], half[b,n,f], int64[b,n,2] -> [p1_new int64[b,n,2]]"""
    _m.require_contiguous(D11=D11, D21=D21, p1=p1)
    if D11.dtype != torch.float16 or D21.dtype != torch.float16:
        # The reference dispatches float/double/half; the SLAM path only ever passes half
        # (matching.py:78-84) and the fp16 accumulate is part of the contract.
        raise RuntimeError("refine_matches: D11/D21 must be float16 (the only dtype the SLAM path uses)")
    _m.require_dtype(p1, torch.int64, "p1")
    b, h, w, f = D11.shape
    n = p1.shape[1]
    p1_new = torch.empty((b, n, 2), dtype=p1.dtype, device=p1.device)   # fully written by the kernel
    rc = _m.lib().mslam_refine_matches(
        _m.ptr(D11), _m.ptr(D21), _m.ptr(p1), _m.ptr(p1_new), b, h, w, n, f, int(radius), int(dilation_max),
        _m.stream_ptr(),
    )
    _m.check(rc, "refine_matches")
    return [p1_new]


# ------------------------------------------------------------------------------------------------
# Gauss-Newton backend
# ------------------------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only per-device scratch buffer (the reference allocates Hs/gs/dx per call with
    torch::zeros, gn_kernels.cu:1173-1174; a cached buffer keeps the call graph-capturable)."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _gn_common_checks(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q):
    _m.require_contiguous(Twc=Twc, Xs=Xs, Cs=Cs, ii=ii, jj=jj, idx_ii2jj=idx_ii2jj, valid_match=valid_match, Q=Q)
    for name, t in (("Twc", Twc), ("Xs", Xs), ("Cs", Cs), ("Q", Q)):
        _m.require_dtype(t, torch.float32, name)
    for name, t in (("ii", ii), ("jj", jj), ("idx_ii2jj", idx_ii2jj)):
        _m.require_dtype(t, torch.int64, name)
    _m.require_dtype(valid_match, torch.bool, "valid_match")
    P, HW = Xs.shape[0], Xs.shape[1]
    E = ii.shape[0]
    if Twc.shape[0] != P:
        raise RuntimeError(f"Twc has {Twc.shape[0]} poses but Xs has {P}")
    return P, HW, E


def gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray, sigma_dist, C_thresh,
                      Q_thresh, max_iter, delta_thresh):
    """gn.cpp:28-52 / gn_kernels.cu:1140-1228.  Mutates Twc in place (rows >= 1); returns [dx]."""
    P, HW, E = _gn_common_checks(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    dx = torch.zeros((max(P - 1, 0), 7), dtype=torch.float32, device=Twc.device)
    nbytes = _m.lib().mslam_gn_workspace_bytes(P, E, HW, E)
    ws = _workspace(nbytes, Twc.device)
    rc = _m.lib().mslam_gauss_newton_rays(
        _m.ptr(Twc), _m.ptr(Xs), _m.ptr(Cs), _m.ptr(ii), _m.ptr(jj), _m.ptr(idx_ii2jj), _m.ptr(valid_match),
        _m.ptr(Q), P, HW, E, float(sigma_ray), float(sigma_dist), float(C_thresh), float(Q_thresh),
        int(max_iter), float(delta_thresh), _m.ptr(dx), _m.ptr(ws), ws.numel(), _m.stream_ptr(),
    )
    _m.check(rc, "gauss_newton_rays")
    return [dx]


def gauss_newton_calib(Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, height, width, pixel_border, z_eps,
                       sigma_pixel, sigma_depth, C_thresh, Q_thresh, max_iter, delta_thresh):
    """gn.cpp:54-82 / gn_kernels.cu:1546-1638."""
    P, HW, E = _gn_common_checks(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    _m.require_contiguous(K=K)
    _m.require_dtype(K, torch.float32, "K")
    dx = torch.zeros((max(P - 1, 0), 7), dtype=torch.float32, device=Twc.device)
    nbytes = _m.lib().mslam_gn_workspace_bytes(P, E, HW, E)
    ws = _workspace(nbytes, Twc.device)
    rc = _m.lib().mslam_gauss_newton_calib(
        _m.ptr(Twc), _m.ptr(Xs), _m.ptr(Cs), _m.ptr(K), _m.ptr(ii), _m.ptr(jj), _m.ptr(idx_ii2jj),
        _m.ptr(valid_match), _m.ptr(Q), P, HW, E, int(height), int(width), int(pixel_border), float(z_eps),
        float(sigma_pixel), float(sigma_depth), float(C_thresh), float(Q_thresh), int(max_iter),
        float(delta_thresh), _m.ptr(dx), _m.ptr(ws), ws.numel(), _m.stream_ptr(),
    )
    _m.check(rc, "gauss_newton_calib")
    return [dx]


def gauss_newton_points(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_point, C_thresh, Q_thresh,
                        max_iter, delta_thresh):
    """gn.cpp:3-26 / gn_kernels.cu:725-811 (exported by the reference, never called from its Python)."""
    P, HW, E = _gn_common_checks(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    dx = torch.zeros((max(P - 1, 0), 7), dtype=torch.float32, device=Twc.device)
    nbytes = _m.lib().mslam_gn_workspace_bytes(P, E, HW, E)
    ws = _workspace(nbytes, Twc.device)
    rc = _m.lib().mslam_gauss_newton_points(
        _m.ptr(Twc), _m.ptr(Xs), _m.ptr(Cs), _m.ptr(ii), _m.ptr(jj), _m.ptr(idx_ii2jj), _m.ptr(valid_match),
        _m.ptr(Q), P, HW, E, float(sigma_point), float(C_thresh), float(Q_thresh), int(max_iter),
        float(delta_thresh), _m.ptr(dx), _m.ptr(ws), ws.numel(), _m.stream_ptr(),
    )
    _m.check(rc, "gauss_newton_points")
    return [dx]


# --- opened-up loop (multi-GPU factor graph; see include/mslam_hip.h) ---------------------------
_KIND = {"rays": 0, "calib": 1, "points": 2}


def gn_blocks(kind, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, sigma_a, sigma_b, C_thresh, Q_thresh,
              height=0, width=0, pixel_border=0, z_eps=0.0, edge_begin=0, edge_count=None, Hs=None, gs=None):
    """One accumulation pass: returns (Hs f32[4,E,7,7], gs f32[2,E,7]) in the reference's block
    layout (gn_kernels.cu:1120-1133) for edges [edge_begin, edge_begin+edge_count)."""
    P, HW = Xs.shape[0], Xs.shape[1]
    E = ii.shape[0]
    edge_count = E - edge_begin if edge_count is None else edge_count
    dev = Twc.device
    if Hs is None:
        Hs = torch.zeros((4, E, 7, 7), dtype=torch.float32, device=dev)
        gs = torch.zeros((2, E, 7), dtype=torch.float32, device=dev)
    ws = _workspace(_m.lib().mslam_gn_workspace_bytes(P, E, HW, int(edge_count)), dev)
    L = _m.lib()
    _m.check(L.mslam_gn_begin(_m.ptr(ii), _m.ptr(jj), P, E, HW, _m.ptr(ws), ws.numel(), _m.stream_ptr()), "gn_begin")
    rc = L.mslam_gn_compact(
        _m.ptr(Xs), _m.ptr(Cs), _m.ptr(idx_ii2jj), _m.ptr(valid_match), _m.ptr(Q), P, HW, E, int(edge_begin),
        int(edge_count), float(C_thresh), float(Q_thresh), _m.ptr(ws), ws.numel(), _m.stream_ptr())
    _m.check(rc, "gn_compact")
    rc = L.mslam_gn_accumulate(
        _KIND[kind], _m.ptr(Twc), _m.ptr(K), P, HW, E, int(edge_begin), int(edge_count), float(sigma_a),
        float(sigma_b), int(height), int(width), int(pixel_border), float(z_eps), _m.ptr(Hs), _m.ptr(gs),
        _m.ptr(ws), ws.numel(), _m.stream_ptr(),
    )
    _m.check(rc, "gn_accumulate")
    return Hs, gs
