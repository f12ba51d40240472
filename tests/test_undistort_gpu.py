"""The remap kernel (csrc/undistort.hip: cv2.remap INTER_LINEAR for 8-bit images, dataloader.py:495-496) against the NumPy
restatement of the same fixed-point arithmetic - bit for bit - and through the calibrated TUM reader.  OpenCV is absent:
parity with cv2 is unpinned (mast3r_slam/undistort.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_remap_kernel_equals_the_integer_restatement(device):
    from mast3r_slam import undistort as ud

    rng = np.random.default_rng(3)
    H, W = 480, 640
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    ident = ud.DeviceRemap(jj, ii, device)(img)
    assert np.array_equal(ident, img)
    # arbitrary positions incl. exact .5 ties of the 1/32 rounding, negative and beyond-the-edge coordinates
    mapx = (jj + rng.uniform(-40, 40, (H, W))).astype(np.float32)
    mapy = (ii + rng.uniform(-40, 40, (H, W))).astype(np.float32)
    mapx[::7, ::5] = np.round(mapx[::7, ::5] * 32) / 32 + 1 / 64
    mapy[::3, ::11] = np.round(mapy[::3, ::11] * 32) / 32 - 1 / 64
    out = ud.DeviceRemap(mapx, mapy, device)(img)
    assert np.array_equal(out, ud.remap_reference(img, mapx, mapy))
    gray = ud.DeviceRemap(mapx, mapy, device)(img[..., 0].copy())
    assert gray.shape == (H, W) and np.array_equal(gray, out[..., 0])
    # a smaller destination than the source
    sub = ud.DeviceRemap(mapx[:100, :200], mapy[:100, :200], device)(img)
    assert np.array_equal(sub, out[:100, :200])


def test_calibrated_tum_reader_undistorts(device, tmp_path, monkeypatch):
    """TUMDataset with use_calib (dataloader.py:77-92): intrinsics from the sequence name, undistortion maps, remap per
    image.  A checkerboard photographed through the lens model comes out straight: rows of the undistorted image that
    cross a horizontal edge of the scene are constant along x."""
    import PIL.Image

    from mast3r_slam import undistort as ud
    from mast3r_slam.config import config
    from mast3r_slam.dataloader import TUMDataset

    W, H = 640, 480
    K = np.array([[517.3, 0.0, 318.6], [0.0, 516.5, 255.3], [0.0, 0.0, 1.0]])
    dist = [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]
    # scene: horizontal stripes in NORMALISED (ideal pinhole) coordinates; the camera image is their distorted view
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    xy = ud.undistort_points(np.stack((jj.ravel(), ii.ravel()), 1), K, dist, iters=40)
    stripes = ((np.floor(xy[:, 1] * 12.0) % 2) * 255).astype(np.uint8).reshape(H, W)
    seq = tmp_path / "tum" / "rgbd_dataset_freiburg1_room"
    (seq / "rgb").mkdir(parents=True)
    PIL.Image.fromarray(np.stack([stripes] * 3, -1)).save(seq / "rgb" / "1.000000.png")
    (seq / "rgb.txt").write_text("# color images\n# file\n# timestamp filename\n1.000000 rgb/1.000000.png\n")
    monkeypatch.setitem(config, "use_calib", True)
    ds = TUMDataset(str(seq))
    assert ds.camera_intrinsics is not None and ds.use_calibration
    ts, img = ds[0]
    assert img.shape == (H, W, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
    Kn = ds.camera_intrinsics.K
    # in the undistorted image a scene row y = const is an image row: away from the stripe edges every row is constant
    rows_y = (np.arange(H) - Kn[1, 2]) / Kn[1, 1] * 12.0
    clean = np.abs(rows_y - np.round(rows_y)) > 0.15              # rows at least 0.15 of a stripe away from an edge
    spread = img[clean, 40:-40, 0].max(axis=1) - img[clean, 40:-40, 0].min(axis=1)
    assert clean.sum() > 200 and (spread < 0.02).mean() > 0.97, (clean.sum(), (spread < 0.02).mean())
