"""Dataset readers (mast3r_slam/dataloader.py) against the reference module's own readers run on the same
directory tree (tests/golden/dataloader.npz; tree rebuilt from the fixture's images by tests/golden/dataset_tree.py)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from dataset_tree import build_dataset_tree  # noqa: E402


@pytest.fixture(scope="module")
def tree(tmp_path_factory, golden_dir):
    fx = np.load(os.path.join(golden_dir, "dataloader.npz"))
    root = tmp_path_factory.mktemp("datasets")
    layout = build_dataset_tree(root, fx["imgs"])
    assert layout == json.loads(str(fx["layout"]))
    return fx, str(root), layout


@pytest.mark.parametrize("cls", ["TUMDataset", "SevenScenesDataset", "RGBFiles", "ETH3DDataset", "ReplicaDataset"])
def test_reader_matches_reference(tree, cls):
    from mast3r_slam import dataloader as dl

    fx, root, layout = tree
    ds = dl.load_dataset(os.path.join(root, layout[cls]))
    assert type(ds).__name__ == cls
    assert len(ds) == int(fx[f"{cls}_len"])
    assert [str(t) for t in ds.timestamps] == list(fx[f"{cls}_timestamps"])
    assert [os.path.relpath(str(f), root) for f in ds.rgb_files] == list(fx[f"{cls}_files"])
    t0, im0 = ds[0]
    assert str(t0) == str(fx[f"{cls}_t0"]) and im0.dtype == np.float32
    np.testing.assert_array_equal(im0, fx[f"{cls}_img0"])
    np.testing.assert_array_equal(ds[len(ds) - 1][1], fx[f"{cls}_imgN"])
    shp, raw = ds.get_img_shape()
    assert [*shp, *raw] == fx[f"{cls}_shape"].tolist()
    assert [ds.has_calib(), ds.use_calibration, ds.save_results] == fx[f"{cls}_flags"].tolist()
    ds.subsample(2)
    assert [os.path.relpath(str(f), root) for f in ds.rgb_files] == list(fx[f"{cls}_sub_files"])


def test_calibrated_dataset_builds_its_maps_and_live_sources_fail_loudly(tree, monkeypatch):
    """use_calib: the TUM reader builds the optimal new camera matrix and the undistortion maps (mast3r_slam/undistort.py,
    OpenCV's published algorithms - cv2 itself is absent, parity unpinned; the remap kernel needs a device, see
    tests/test_undistort_gpu.py).  Live / video sources still raise."""
    from mast3r_slam import dataloader as dl
    from mast3r_slam.config import config

    fx, root, layout = tree
    monkeypatch.setitem(config, "use_calib", True)
    ds = dl.load_dataset(os.path.join(root, layout["TUMDataset"]))
    intr = ds.camera_intrinsics
    assert ds.has_calib() and intr.mapx.shape == (480, 640) and intr.mapx.dtype == np.float32
    assert intr.K[0, 2] == 319.5 and intr.K[1, 2] == 239.5          # config.dataset.center_principle_point
    assert 0.8 * intr.K_orig[0, 0] < intr.K[0, 0] < 1.3 * intr.K_orig[0, 0]
    monkeypatch.setitem(config, "use_calib", False)
    for path in ("realsense", "webcam", "clip.mp4"):
        with pytest.raises(NotImplementedError):
            dl.load_dataset(path)


def test_k_frame_and_synthetic_room():
    from mast3r_slam import dataloader as dl

    K = np.array([[517.3, 0.0, 318.6], [0.0, 516.5, 255.3], [0.0, 0.0, 1.0]])
    intr = dl.Intrinsics(512, 640, 480, K, K.copy(), np.zeros(4), None, None)     # 640x480 -> 512x384: scale 1.25, no crop
    np.testing.assert_allclose(intr.K_frame, [[517.3 / 1.25, 0, 318.6 / 1.25], [0, 516.5 / 1.25, 255.3 / 1.25], [0, 0, 1]])
    ds = dl.SyntheticRoomDataset(n_frames=6, stride=4, h=48, w=64)
    t, img = ds[3]
    assert len(ds) == 6 and img.shape == (48, 64, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
    assert abs(float(t) - 0.1) < 1e-6
    assert dl.natsorted(["f10.png", "f2.png", "F1.png"]) == ["F1.png", "f2.png", "f10.png"]
