"""Edge cases and error behaviour through the product API (-m gpu): empty / degenerate inputs, overflow reporting,
and the reference's error contract (non-contiguous input -> RuntimeError("<name> must be contiguous"), gn.h:5)."""
import numpy as np
import pytest
import torch

import oracle
from mast3r_slam import synthetic

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_gn_edge_without_valid_matches_and_two_pose_graph(device):
    """An edge whose matches are all invalid contributes exactly zero blocks; the smallest graph (two poses, one
    edge pair) solves; both as in the oracle."""
    import mast3r_slam_backends as be

    g = synthetic.make_graph(n_kf=2, h=24, w=32, seed=2, extra_edges=0)
    d = {k: _t(v, device) for k, v in g.items() if isinstance(v, np.ndarray)}
    vm = d["valid_match"].clone()
    vm[1] = False
    Hs, gs = be.gn_blocks("rays", d["Twc"], d["Xs"], d["Cs"], None, d["ii"], d["jj"], d["idx_ii2jj"], vm, d["Q"], 0.003, 10.0,
                          0.0, 1.5)
    assert float(Hs[:, 1].abs().max()) == 0.0 and float(gs[:, 1].abs().max()) == 0.0 and float(Hs[:, 0].abs().max()) > 0
    Twc = d["Twc"].clone()
    dx = be.gauss_newton_rays(Twc, d["Xs"], d["Cs"], d["ii"], d["jj"], d["idx_ii2jj"], d["valid_match"], d["Q"], 0.003, 10.0,
                              0.0, 1.5, 3, 1e-8)[0]
    ref = oracle.gauss_newton("rays", g["Twc"], g["Xs"], g["Cs"], None, g["ii"], g["jj"], g["idx_ii2jj"], g["valid_match"],
                              g["Q"], 0.003, 10.0, 0.0, 1.5, max_iter=3, delta_thresh=1e-8)
    assert dx.shape == (1, 7) and torch.equal(Twc[0], d["Twc"][0])          # first pose is pinned
    np.testing.assert_allclose(Twc.cpu().numpy(), ref[0], atol=2e-4)


def test_gn_rejects_non_contiguous_and_host_tensors(device):
    import mast3r_slam_backends as be

    g = synthetic.make_graph(n_kf=3, h=24, w=32, seed=1)
    d = {k: _t(v, device) for k, v in g.items() if isinstance(v, np.ndarray)}
    args = [d["Twc"], d["Xs"], d["Cs"], d["ii"], d["jj"], d["idx_ii2jj"], d["valid_match"], d["Q"]]
    bad = list(args)
    bad[1] = d["Xs"].transpose(0, 1).contiguous().transpose(0, 1)     # same values, non-contiguous strides
    with pytest.raises(RuntimeError, match="must be contiguous"):
        be.gauss_newton_rays(*bad, 0.003, 10.0, 0.0, 1.5, 2, 1e-3)
    bad = list(args)
    bad[2] = d["Cs"].cpu()
    with pytest.raises(RuntimeError, match="no CPU"):
        be.gauss_newton_rays(*bad, 0.003, 10.0, 0.0, 1.5, 2, 1e-3)


def test_tsdf_degenerate_inputs_and_overflow(device):
    from mast3r_slam.tsdf import TSDFVolume

    vol = TSDFVolume(0.03, 0.12, capacity=1 << 10, device=device)
    assert vol.integrate(np.zeros((0, 3), np.float32), np.zeros(0), np.zeros(3, np.float32)) == 0
    assert vol.stats()["total_voxels"] == 0
    # zero-confidence points fuse nothing (global_volume.py:50-52 skips w <= 0)
    pts = np.random.default_rng(0).uniform(-1, 1, (50, 3)).astype(np.float32) + np.array([0, 0, 3], np.float32)
    assert vol.integrate(pts, np.zeros(50), np.zeros(3, np.float32)) == oracle.TSDFVolume(0.03, 0.12).integrate(
        pts, np.zeros(50), np.zeros(3, np.float32))
    # a table that is too small reports the overflow instead of dropping voxels silently
    big = np.random.default_rng(1).uniform(-3, 3, (4000, 3)).astype(np.float32) + np.array([0, 0, 8], np.float32)
    with pytest.raises(RuntimeError, match="overflow"):
        vol.integrate(big, np.full(4000, 2.0), np.zeros(3, np.float32))
    v, gq, st = vol.query_batch(np.array([[100.0, 100.0, 100.0]], np.float32))   # far from anything: no value
    assert int(st[0]) == 0


def test_local_tsdf_without_confident_points(device):
    """_build_tsdf_robust with every confidence below min_confidence leaves the block untouched (tsdf 1, weight 0),
    and the ray cast over it reports no hits."""
    from lietorch_hip import Sim3
    from mast3r_slam.tsdf_refine import TSDFRefiner

    cfg = dict(voxel_size=0.02, trunc_dist=0.08, max_grid_dim=64, roi_size=0.4, ray_samples=64, max_displacement=0.015,
               min_weight_threshold=0.01, confidence_boost=0.08, confidence_max=1.3, min_hit_rate=0.05, min_confidence=0.2)
    ref = TSDFRefiner(cfg, None, None, device)
    T = synthetic.camera_pose(0)
    X = synthetic.render_pointmap(T, 48, 64).reshape(-1, 3).astype(np.float32)
    C = np.full(X.shape[0], 0.05, np.float32)
    mn, mx = _t(X[:200].min(0) - 0.02, device), _t(X[:200].max(0) + 0.02, device)
    tsdf, w = ref._build_tsdf_robust(_t(X, device), _t(C, device), None, mn, mx, 48, 64, Sim3.Identity(1, device=device))
    assert float(w.abs().max()) == 0.0 and float((tsdf - 1.0).abs().max()) == 0.0
    mask = torch.zeros(X.shape[0], dtype=torch.bool, device=device)
    mask[:200] = True
    Xr, hits = ref._extract_surface_safe(tsdf, mn, mx, None, mask, 48, 64, _t(X, device), order=torch.arange(100))
    assert int(hits.sum()) == 0 and torch.equal(Xr, _t(X, device))
