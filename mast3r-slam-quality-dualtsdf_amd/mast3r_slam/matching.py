"""Mirror of the reference's mast3r_slam/matching.py (lines 8-90): same function names, arguments
and return values; every numeric step is a HIP kernel from libmslam_hip.so.

Differences in *how* (not what):
  * prep_for_iter_proj is ONE fused kernel (normalise + 3x3 gradients + concat + normalise X21 +
    p_init) instead of ~10 torch ops (matching.py:25-49, image.py:5-38).
  * the occlusion test and p.long() are one kernel (matching.py:68-76); pixel_to_lin one kernel.
"""
import torch

import mast3r_slam_backends
import mslam_hip as _m
from mast3r_slam.config import config


def match(X11, X21, D11, D21, idx_1_to_2_init=None):
    idx_1_to_2, valid_match2 = match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init)
    return idx_1_to_2, valid_match2


def pixel_to_lin(p1, w):
    """matching.py:13-15 — idx = u + w*v for an int64 (..., 2) pixel tensor."""
    p1 = p1.contiguous()
    lead = p1.shape[:-1]
    n = 1
    for s in lead:
        n *= s
    idx = torch.empty(lead, dtype=torch.int64, device=p1.device)
    rc = _m.lib().mslam_pixel_to_lin(_m.ptr(p1), _m.ptr(idx), 1, n, int(w), _m.stream_ptr())
    _m.check(rc, "pixel_to_lin")
    return idx


def lin_to_pixel(idx_1_to_2, w):
    """matching.py:18-22 (index plumbing; integer ops on the device tensor)."""
    u = idx_1_to_2 % w
    v = idx_1_to_2 // w
    return torch.stack((u, v), dim=-1)


def prep_for_iter_proj(X11, X21, idx_1_to_2_init):
    """matching.py:25-49.  X11,X21 f32[b,h,w,3] -> (rays_with_grad f32[b,h,w,9],
    pts3d_norm f32[b,hw,3], p_init f32[b,hw,2])."""
    b, h, w, _ = X11.shape
    X11 = X11.contiguous()
    X21 = X21.contiguous()
    dev = X11.device
    rays = torch.empty((b, h, w, 9), dtype=torch.float32, device=dev)
    pts = torch.empty((b, h * w, 3), dtype=torch.float32, device=dev)
    p_init = torch.empty((b, h * w, 2), dtype=torch.float32, device=dev)
    if idx_1_to_2_init is not None:
        idx_1_to_2_init = idx_1_to_2_init.contiguous()
        _m.require_dtype(idx_1_to_2_init, torch.int64, "idx_1_to_2_init")
    rc = _m.lib().mslam_prep_iter_proj(
        _m.ptr(X11), _m.ptr(X21), _m.ptr(idx_1_to_2_init), _m.ptr(rays), _m.ptr(pts), _m.ptr(p_init),
        b, h, w, _m.stream_ptr(),
    )
    _m.check(rc, "prep_for_iter_proj")
    return rays, pts, p_init


def match_iterative_proj(X11, X21, D11, D21, idx_1_to_2_init=None):
    """matching.py:52-90."""
    cfg = config["matching"]
    b, h, w = X21.shape[:3]
    X11 = X11.contiguous()
    X21 = X21.contiguous()

    rays_with_grad_img, pts3d_norm, p_init = prep_for_iter_proj(X11, X21, idx_1_to_2_init)
    p1f, valid_proj2 = mast3r_slam_backends.iter_proj(
        rays_with_grad_img, pts3d_norm, p_init, cfg["max_iter"], cfg["lambda_init"], cfg["convergence_thresh"]
    )

    # p1 = p1.long(); occlusion test on 3-D distance (matching.py:68-76)
    p1 = torch.empty((b, h * w, 2), dtype=torch.int64, device=X11.device)
    rc = _m.lib().mslam_match_occlusion(
        _m.ptr(X11), _m.ptr(X21), _m.ptr(p1f), _m.ptr(p1), _m.ptr(valid_proj2), b, h, w,
        float(cfg["dist_thresh"]), _m.stream_ptr(),
    )
    _m.check(rc, "match_occlusion")

    if cfg["radius"] > 0:
        (p1,) = mast3r_slam_backends.refine_matches(
            D11.half().contiguous(), D21.reshape(b, h * w, -1).half().contiguous(), p1,
            cfg["radius"], cfg["dilation_max"],
        )

    idx_1_to_2 = pixel_to_lin(p1, w)
    return idx_1_to_2, valid_proj2.unsqueeze(-1)
