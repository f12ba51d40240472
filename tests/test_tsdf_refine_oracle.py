"""CPU test (-m "not gpu"): local dense-block TSDF oracle vs the fixture produced by running the
reference's tsdf_refine.py (grid indices exact, values to fp32 rounding)."""
import os

import numpy as np
import pytest

from oracle import tsdf_refine_py as TR
from mast3r_slam import synthetic


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "tsdf_refine.npz"))


@pytest.mark.parametrize("case", ["A", "B"])
def test_build_and_raycast_match_reference(fx, case):
    T = fx[f"{case}_pose"].astype(np.float64)
    Xw = synthetic.sim3_act(T, fx["X"].astype(np.float64)).astype(np.float32)   # same duck-typed act as the fixture
    tsdf, weights = TR.build_tsdf(Xw, fx["C"], T[:3].astype(np.float32), fx[f"{case}_xyz_min"], fx[f"{case}_xyz_max"],
                                  linspace="torch")
    assert tsdf.shape == fx[f"{case}_tsdf"].shape
    np.testing.assert_array_equal(weights > 0, fx[f"{case}_weights"] > 0)        # touched voxel set: exact
    np.testing.assert_allclose(weights, fx[f"{case}_weights"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(tsdf, fx[f"{case}_tsdf"], rtol=0, atol=2e-6)
    Xr, hits = TR.extract_surface(fx[f"{case}_tsdf"], fx[f"{case}_xyz_min"], fx[f"{case}_xyz_max"], fx[f"{case}_mask"],
                                  fx["X"], fx[f"{case}_perm"], linspace="torch")
    np.testing.assert_array_equal(hits, fx[f"{case}_hits"])
    np.testing.assert_allclose(Xr, fx[f"{case}_X_refined"], rtol=0, atol=1e-6)
    if case == "A":
        assert hits.sum() > 0


def test_scalar_linspace_is_one_ulp_from_torch_cpu(fx):
    """The device form of linspace (per-element formula) and torch's vectorised CPU kernel agree to 1 ulp."""
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, n = rng.uniform(0.1, 3.0), int(rng.integers(1, 33))
        x, y = TR.linspace_f32(a, a + 0.32, n, "torch"), TR.linspace_f32(a, a + 0.32, n, "scalar")
        assert np.abs(x.view(np.int32) - y.view(np.int32)).max() <= 1
