"""CPU tests (-m "not gpu"): the matching oracle against the reference-generated golden fixture,
against known-answer geometry, and against its own documented conventions."""
import os

import numpy as np
import pytest

import oracle
from oracle import matching_py
from mast3r_slam import synthetic


def test_prep_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "prep_iter_proj.npz"))
    rays, pts, p_init = matching_py.prep_for_iter_proj(g["X11"], g["X21"])
    # torch's conv/normalise summation order is unspecified -> fp32 rounding tolerance
    np.testing.assert_allclose(rays, g["rays_with_grad"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(pts, g["pts3d_norm"], rtol=0, atol=2e-7)
    assert p_init[0, 65].tolist() == [1.0, 1.0]


def _pair(h=48, w=64, kj=4):
    return synthetic.make_pair(0, kj, h=h, w=w, seed=1, noise=0.0)


def test_iter_proj_recovers_known_projection():
    """Known answer: X21 are view-j points in frame i, so the LM must land where the pinhole
    model projects them (inside the image, where the room surface is smooth)."""
    pr = _pair()
    h, w = pr["X11"].shape[:2]
    rays, pts, p0 = matching_py.prep_for_iter_proj(pr["X11"][None], pr["X21"][None])
    p, conv = oracle.iter_proj(rays, pts, p0, 10, 1e-8, 1e-6)
    K = pr["K"]
    X = pr["X21"].reshape(-1, 3).astype(np.float64)
    u = K[0, 0] * X[:, 0] / X[:, 2] + K[0, 2]
    v = K[1, 1] * X[:, 1] / X[:, 2] + K[1, 2]
    inside = (u > 2) & (u < w - 3) & (v > 2) & (v < h - 3) & (X[:, 2] > 0.1) & conv[0]
    assert inside.sum() > 0.3 * h * w
    err = np.hypot(p[0, :, 0] - u, p[0, :, 1] - v)[inside]
    # bilinear interpolation of unit rays across wall edges limits accuracy; median must be tiny
    assert np.median(err) < 0.05
    assert np.percentile(err, 90) < 0.5


def test_iter_proj_edge_cases():
    pr = _pair(16, 16)
    rays, pts, p0 = matching_py.prep_for_iter_proj(pr["X11"][None], pr["X21"][None])
    # zero iterations: clamped init comes back, converged all False
    p, conv = oracle.iter_proj(rays, pts, p0, 0, 1e-8, 1e-6)
    assert not conv.any()
    np.testing.assert_array_equal(p, np.clip(p0, 1, 14))
    # empty batch
    p, conv = oracle.iter_proj(rays[:0], pts[:0], p0[:0], 10, 1e-8, 1e-6)
    assert p.shape == (0, 256, 2)
    # results stay inside the clamp box
    p, _ = oracle.iter_proj(rays, pts, p0 + 100.0, 10, 1e-8, 1e-6)
    assert p.min() >= 1 and p.max() <= 14


def _desc_inputs(h=32, w=40, seed=0):
    rng = np.random.default_rng(seed)
    D11 = rng.normal(size=(1, h, w, 24)).astype(np.float32)
    D11 /= np.linalg.norm(D11, axis=-1, keepdims=True)
    return D11.astype(np.float16), rng


def test_refine_matches_finds_planted_descriptor():
    D11, rng = _desc_inputs()
    h, w = D11.shape[1:3]
    n = h * w
    tu = rng.integers(0, w, n)
    tv = rng.integers(0, h, n)
    D21 = D11[0, tv, tu][None]  # exact copy of the target pixel's descriptor
    # start within the first dilation level's reach of the target (|offset| multiple of 5, <= 15)
    du = rng.integers(-3, 4, n) * 5
    dv = rng.integers(-3, 4, n) * 5
    p1 = np.stack((np.clip(tu + du, 0, w - 1), np.clip(tv + dv, 0, h - 1)), -1)[None].astype(np.int64)
    reach = (np.abs(p1[0, :, 0] - tu) % 5 == 0) & (np.abs(p1[0, :, 1] - tv) % 5 == 0)
    out = oracle.refine_matches(D11, D21, p1, 3, 5)
    hit = (out[0, :, 0] == tu) & (out[0, :, 1] == tv)
    assert hit[reach].mean() > 0.97  # fp16 near-ties aside, the planted unit-norm copy wins


def test_refine_matches_quirks():
    """Appendix B.1: scores <= 2^-14 never win; borders are skipped; ties keep the first."""
    h, w = 8, 8
    D11 = np.zeros((1, h, w, 24), np.float16)
    D21 = np.zeros((1, h * w, 24), np.float16)
    D21[..., 0] = 1.0
    D11[0, :, :, 0] = np.float16(6.0e-5)  # below half::min -> nobody wins, p1 unchanged
    p1 = np.stack(np.meshgrid(np.arange(w), np.arange(h), indexing="xy"), -1).reshape(1, -1, 2).astype(np.int64)
    out = oracle.refine_matches(D11, D21, p1, 3, 5)
    np.testing.assert_array_equal(out, p1)
    # all-equal winning scores: first candidate in scan order (u outer, v inner) that is inside
    D11[0, :, :, 0] = 0.5
    out = oracle.refine_matches(D11, D21, p1, 1, 1)
    # radius 1, dilation 1: first inside candidate is (max(u-1,0), max(v-1,0))
    np.testing.assert_array_equal(out[0, :, 0], np.maximum(p1[0, :, 0] - 1, 0))
    np.testing.assert_array_equal(out[0, :, 1], np.maximum(p1[0, :, 1] - 1, 0))


def test_refine_matches_half_accumulate_is_sequential():
    """One candidate, so p1_new tells us whether the half-precision sum beat the threshold:
    construct a sum that is > 2^-14 in fp32 but rounds to <= 2^-14 step by step in fp16."""
    D11 = np.zeros((1, 1, 1, 24), np.float16)
    D21 = np.zeros((1, 1, 24), np.float16)
    # 2048 + 1 + 1 ... in half stays 2048 (ulp = 2); products 1*1
    D11[0, 0, 0, 0], D21[0, 0, 0] = 2048.0, 1.0
    D11[0, 0, 0, 1:], D21[0, 0, 1:] = 1.0, 1.0
    from oracle import lib, _p
    import ctypes
    # expose the score through a 2-pixel image: pixel 0 has the big sum, pixel 1 holds 2049
    img = np.zeros((1, 1, 2, 24), np.float16)
    img[0, 0, 0] = D11[0, 0, 0]
    img[0, 0, 1, 0] = 2050.0  # sequential fp16 sum at pixel0 = 2048 < 2050 -> pixel 1 must win
    p1 = np.zeros((1, 1, 2), np.int64)
    out = oracle.refine_matches(img, D21, p1, 1, 1)
    assert out[0, 0].tolist() == [1, 0]


def test_half_conversion_roundtrip_exhaustive():
    """The oracle's software binary16 must agree with numpy for every finite half."""
    bits = np.arange(0, 0x7C00, dtype=np.uint16)
    h = bits.view(np.float16)
    D11 = np.zeros((1, 1, 1, 8), np.float16)
    # sum of (x * 1) over one element == x exactly: exercise via hadd/hmul with identity
    D21 = np.ones((1, 1, 8), np.float16)
    for x in h[:: 97]:
        D11[...] = 0
        D11[0, 0, 0, 0] = x
        img = np.concatenate((D11, D11), 2)
        img[0, 0, 1, 0] = np.nextafter(x, np.float16(np.inf))
        out = oracle.refine_matches(img, D21, np.zeros((1, 1, 2), np.int64), 1, 1)
        if float(np.nextafter(x, np.float16(np.inf))) > 2 ** -14:
            assert out[0, 0, 0] == 1
