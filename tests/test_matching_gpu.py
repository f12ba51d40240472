"""GPU parity tests (-m gpu): HIP matching kernels, called through the C ABI via the drop-in
Python mirror, against the CPU oracle on identical seeded inputs.

Bars:
  iter_proj       p_new bit-exact vs the oracle restatement (both use explicit fmaf + fp64 for the
                  reference's double sub-expressions); `converged` equal.
                  (vs the un-runnable CUDA binary the stated tolerance is 1e-3 px on 99.9 %.)
  refine_matches  int64 indices bit-exact (IEEE half mul/add, sequential k).
  prep            fp32 tolerance 2e-6 vs the reference-generated golden (torch op order).
  occlusion/lin   bool / int64 exact.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import matching_py
from mast3r_slam import synthetic

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _pair_batch(h, w, ks=((0, 4), (10, 13))):
    prs = [synthetic.make_pair(a, b, h=h, w=w, seed=2) for a, b in ks]
    st = lambda k: np.stack([p[k] for p in prs])
    return st("X11"), st("X21"), st("D11"), st("D21")


@pytest.mark.parametrize("h,w", [(48, 64), (384, 512)])
def test_iter_proj_bit_exact(device, h, w):
    import mast3r_slam_backends as be

    X11, X21, _, _ = _pair_batch(h, w)
    rays, pts, p0 = matching_py.prep_for_iter_proj(X11, X21)
    p_ref, c_ref = oracle.iter_proj(rays, pts, p0, 10, 1e-8, 1e-6)
    p, c = be.iter_proj(_t(rays, device), _t(pts, device), _t(p0, device), 10, 1e-8, 1e-6)
    assert p.dtype == torch.float32 and c.dtype == torch.bool
    p, c = p.cpu().numpy(), c.cpu().numpy()
    mism = (p != p_ref).any(-1)
    assert mism.mean() == 0.0, f"{mism.sum()} of {mism.size} points differ, max {np.abs(p - p_ref).max()}"
    np.testing.assert_array_equal(c, c_ref)


def test_iter_proj_init_from_previous_and_edge_cases(device):
    import mast3r_slam_backends as be

    X11, X21, _, _ = _pair_batch(32, 48)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 32 * 48, (2, 32 * 48))
    rays, pts, p0 = matching_py.prep_for_iter_proj(X11, X21, idx)
    for iters in (0, 1, 3):
        p_ref, c_ref = oracle.iter_proj(rays, pts, p0, iters, 1e-8, 1e-6)
        p, c = be.iter_proj(_t(rays, device), _t(pts, device), _t(p0, device), iters, 1e-8, 1e-6)
        np.testing.assert_array_equal(p.cpu().numpy(), p_ref)
        np.testing.assert_array_equal(c.cpu().numpy(), c_ref)
    # empty batch and ragged n (not a multiple of the block size)
    p, c = be.iter_proj(_t(rays[:0], device), _t(pts[:0], device), _t(p0[:0], device), 10, 1e-8, 1e-6)
    assert p.shape == (0, 32 * 48, 2)
    nr = 1000
    p_ref, c_ref = oracle.iter_proj(rays, pts[:, :nr], p0[:, :nr], 10, 1e-8, 1e-6)
    p, c = be.iter_proj(_t(rays, device), _t(pts[:, :nr], device), _t(p0[:, :nr], device), 10, 1e-8, 1e-6)
    np.testing.assert_array_equal(p.cpu().numpy(), p_ref)


@pytest.mark.parametrize("h,w", [(40, 56), (384, 512)])
def test_refine_matches_bit_exact(device, h, w):
    import mast3r_slam_backends as be

    X11, X21, D11, D21 = _pair_batch(h, w)
    rays, pts, p0 = matching_py.prep_for_iter_proj(X11, X21)
    p, _ = oracle.iter_proj(rays, pts, p0, 10, 1e-8, 1e-6)
    p1 = np.trunc(p).astype(np.int64)
    d11 = D11.astype(np.float16)
    d21 = D21.reshape(2, h * w, -1).astype(np.float16)
    ref = oracle.refine_matches(d11, d21, p1, 3, 5)
    (out,) = be.refine_matches(_t(d11, device), _t(d21, device), _t(p1, device), 3, 5)
    assert out.dtype == torch.int64
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # the refinement must actually move points and stay in the image
    assert (ref != p1).any() and ref[..., 0].max() < w and ref[..., 1].max() < h and ref.min() >= 0


def test_refine_matches_generic_fdim_and_quirks(device):
    import mast3r_slam_backends as be

    rng = np.random.default_rng(5)
    for f in (8, 16, 24, 30):
        d11 = rng.normal(size=(1, 20, 24, f)).astype(np.float16)
        d21 = rng.normal(size=(1, 480, f)).astype(np.float16)
        p1 = np.stack((rng.integers(0, 24, 480), rng.integers(0, 20, 480)), -1)[None].astype(np.int64)
        for radius, dil in ((3, 5), (1, 1), (2, 3)):
            ref = oracle.refine_matches(d11, d21, p1, radius, dil)
            (out,) = be.refine_matches(_t(d11, device), _t(d21, device), _t(p1, device), radius, dil)
            np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # all scores below half::min -> unchanged
    d11 = np.full((1, 8, 8, 24), 1e-4, np.float16)
    d21 = np.full((1, 64, 24), 1e-2, np.float16)
    p1 = np.stack(np.meshgrid(np.arange(8), np.arange(8), indexing="xy"), -1).reshape(1, -1, 2).astype(np.int64)
    (out,) = be.refine_matches(_t(d11, device), _t(d21, device), _t(p1, device), 3, 5)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.refine_matches(d11, d21, p1, 3, 5))


def test_prep_against_reference_golden(device, golden_dir):
    from mast3r_slam import matching

    g = np.load(os.path.join(golden_dir, "prep_iter_proj.npz"))
    rays, pts, p0 = matching.prep_for_iter_proj(_t(g["X11"], device), _t(g["X21"], device), None)
    np.testing.assert_allclose(rays.cpu().numpy(), g["rays_with_grad"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(pts.cpu().numpy(), g["pts3d_norm"], rtol=0, atol=2e-7)
    _, _, p0_ref = matching_py.prep_for_iter_proj(g["X11"], g["X21"])
    np.testing.assert_array_equal(p0.cpu().numpy(), p0_ref)


@pytest.mark.parametrize("h,w", [(30, 44), (384, 512)])
def test_full_match_pipeline(device, h, w):
    """matching.match end to end == oracle composition (indices / flags exact where the
    float inputs to the integer steps are identical; prep differs by fp32 rounding, so the
    comparison feeds the HIP prep output into the oracle)."""
    from mast3r_slam import matching
    from mast3r_slam.config import config

    X11, X21, D11, D21 = _pair_batch(h, w)
    idx, valid = matching.match(_t(X11, device), _t(X21, device), _t(D11, device), _t(D21, device))
    assert idx.shape == (2, h * w) and idx.dtype == torch.int64
    assert valid.shape == (2, h * w, 1) and valid.dtype == torch.bool

    rays, pts, p0 = matching.prep_for_iter_proj(_t(X11, device), _t(X21, device), None)
    cfg = config["matching"]
    p, conv = oracle.iter_proj(rays.cpu().numpy(), pts.cpu().numpy(), p0.cpu().numpy(),
                               cfg["max_iter"], cfg["lambda_init"], cfg["convergence_thresh"])
    p1, v = matching_py.occlusion_and_trunc(X11, X21, p, conv, cfg["dist_thresh"])
    p1 = oracle.refine_matches(D11.astype(np.float16), D21.reshape(2, h * w, -1).astype(np.float16), p1,
                               cfg["radius"], cfg["dilation_max"])
    np.testing.assert_array_equal(idx.cpu().numpy(), matching_py.pixel_to_lin(p1, w))
    np.testing.assert_array_equal(valid.cpu().numpy()[..., 0], v)
    assert v.mean() > 0.2  # the synthetic pair really overlaps
