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
    p_new = torch.zeros((b, n, 2), dtype=p_init.dtype, device=p_init.device)
    converged = torch.zeros((b, n), dtype=torch.bool, device=p_init.device)
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
    p1_new = torch.zeros((b, n, 2), dtype=p1.dtype, device=p1.device)
    rc = _m.lib().mslam_refine_matches(
        _m.ptr(D11), _m.ptr(D21), _m.ptr(p1), _m.ptr(p1_new), b, h, w, n, f, int(radius), int(dilation_max),
        _m.stream_ptr(),
    )
    _m.check(rc, "refine_matches")
    return [p1_new]
