"""TSDFRefiner block selection, clustering and sliding-window scheduling (tsdf_refine.py:246-601) against
tests/golden/refine_schedule.npz, recorded from the reference class itself (tests/golden/make_golden.py
refine_schedule).  Host logic only: runs without a GPU."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "refine_schedule.npz"))


BASE = dict(enabled=True, window_size=5, voxel_size=0.02, trunc_dist=0.08, max_grid_dim=64, roi_size=0.4, ray_samples=64,
            max_displacement=0.015, min_weight_threshold=0.01, confidence_boost=0.08, confidence_max=1.3, min_hit_rate=0.05,
            max_rois_per_kf=3, min_confidence=0.2, max_pending_tasks=50)


class _KF:
    pass


class _Store:
    def __init__(self, fx, n):
        H, W = int(fx["H"]), int(fx["W"])
        self.kfs = []
        for k in range(fx["X"].shape[0]):
            kf = _KF()
            kf.frame_id, kf.img_shape = 10 * k, torch.tensor([[H, W]])
            kf.X_canon, kf.C = torch.from_numpy(fx["X"][k].copy()), torch.from_numpy(fx["C"][k].copy())
            self.kfs.append(kf)
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.kfs[i]


class _Quality:
    def __init__(self, fx):
        self.grids = fx["quality_grids"]

    def get(self, frame_id):
        if frame_id == 30:
            return None
        return {"priority": self.grids[frame_id // 10], "patch_size": 16, "kf_id": frame_id}

    def poll(self):
        pass


def _check_blocks(fx, prefix, blocks):
    assert len(blocks) == int(fx[prefix + "_n"])
    for i, b in enumerate(blocks):
        np.testing.assert_array_equal([b.kf_id, b.block_id], fx[f"{prefix}_{i}_ids"])
        np.testing.assert_array_equal(np.array(b.patch_indices, np.int64).reshape(-1, 2), fx[f"{prefix}_{i}_patches"])
        np.testing.assert_array_equal(np.flatnonzero(b.pixel_mask.numpy()), fx[f"{prefix}_{i}_mask"])
        np.testing.assert_allclose([b.depth_median, b.priority, b.depth_variance], fx[f"{prefix}_{i}_vals"], rtol=1e-6,
                                   atol=1e-9)


def test_block_selection_and_clustering(fx):
    from mast3r_slam.tsdf_refine import TSDFRefiner

    n = fx["X"].shape[0]
    for name, over in (("select_single", {}), ("select_cluster", dict(max_block_edge=2, z_rel=0.5))):
        ref = TSDFRefiner(dict(BASE, **over), _Store(fx, n), None, "cpu")
        _check_blocks(fx, name, ref._select_blocks_enhanced(4, {"priority": torch.from_numpy(fx["priority"]), "patch_size": 16}))


def test_confidence_fallback(fx):
    """No quality result: priority = 0.3 - C on 0.05 < C < 0.3, read with patch_size 16 although it is a per-pixel map
    (the reference's quirk: only pixels of the top-left H/16 x W/16 corner can name a valid patch)."""
    from mast3r_slam.tsdf_refine import TSDFRefiner

    n = fx["X"].shape[0]
    ref = TSDFRefiner(dict(BASE), _Store(fx, n), None, "cpu")
    assert ref._schedule_refinement(2) == bool(fx["fallback_ok"])
    items = list(ref.queue.queue)
    _check_blocks(fx, "fallback", [b for _, b in items])
    assert ref._schedule_refinement(3) == bool(fx["fallback_fail_ok"])
    assert ref._schedule_refinement(5) == bool(fx["fallback5_ok"])
    _check_blocks(fx, "fallback5", [b for _, b in list(ref.queue.queue)[len(items):]])


def test_sliding_window_run_and_final_pass(fx):
    from mast3r_slam.tsdf_refine import TSDFRefiner

    n = fx["X"].shape[0]
    store = _Store(fx, 0)
    ref = TSDFRefiner(dict(BASE), store, _Quality(fx), "cpu")
    for cur in range(n):
        store.n = cur + 1
        ref.maybe_schedule_sliding_window(cur)
        got = np.array([[k.kf_id, k.block_id] for k, _ in ref.queue.queue], np.int64).reshape(-1, 2)
        np.testing.assert_array_equal(got, fx[f"trace_{cur}"], err_msg=f"after keyframe {cur}")
    ref.schedule_final_pass(n - 1)
    got = np.array([[k.kf_id, k.block_id] for k, _ in ref.queue.queue], np.int64).reshape(-1, 2)
    np.testing.assert_array_equal(got, fx[f"trace_{n}"])
    assert ref.is_alive() and ref.queue.qsize() == len(got) and ref.stats["total_blocks"] == 0
