"""Undistortion map construction (mast3r_slam/undistort.py, rebuilt from OpenCV's published algorithms for
dataloader.py:476-516; OpenCV is absent, parity with cv2 is UNPINNED): the properties any correct implementation has."""
import numpy as np
import pytest

from mast3r_slam import undistort as ud

W, H = 640, 480
K = np.array([[517.3, 0.0, 318.6], [0.0, 516.5, 255.3], [0.0, 0.0, 1.0]])
TUM_FR1 = [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]          # dataloader.py: TUM freiburg1 calibration


@pytest.mark.parametrize("center", [False, True])
def test_zero_distortion_is_the_identity(center):
    Kc = K.copy()
    if center:
        Kc[0, 2], Kc[1, 2] = (W - 1) / 2, (H - 1) / 2
    Kn = ud.get_optimal_new_camera_matrix(Kc, np.zeros(4), (W, H), 0, (W, H), center_principal_point=center)
    np.testing.assert_allclose(Kn, Kc, rtol=0, atol=1e-9)
    mapx, mapy = ud.init_undistort_rectify_map(Kc, np.zeros(4), Kn, (W, H))
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    np.testing.assert_allclose(mapx, jj, atol=1e-4)
    np.testing.assert_allclose(mapy, ii, atol=1e-4)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(ud.remap_reference(img, jj.astype(np.float32), ii.astype(np.float32)), img)


def test_distort_then_undistort_returns_the_grid():
    jj, ii = np.meshgrid(np.linspace(0, W - 1, 33), np.linspace(0, H - 1, 25))
    px = np.stack((jj.ravel(), ii.ravel()), 1)
    xy = np.stack(((px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1]), 1)
    d = ud.distort_points(xy, TUM_FR1)
    dpx = np.stack((d[:, 0] * K[0, 0] + K[0, 2], d[:, 1] * K[1, 1] + K[1, 2]), 1)
    back = ud.undistort_points(dpx, K, TUM_FR1, P=K)                       # OpenCV's default: 5 fixed-point iterations
    centre = (np.abs(px[:, 0] - W / 2) < 0.4 * W) & (np.abs(px[:, 1] - H / 2) < 0.4 * H)
    assert np.abs(back - px)[centre].max() < 0.1, np.abs(back - px)[centre].max()
    assert np.abs(back - px).max() < 1.5                                     # the extreme corners of this strong lens
    err = np.abs(ud.undistort_points(dpx, K, TUM_FR1, P=K, iters=60) - px).max()
    assert err < 1e-3, err


@pytest.mark.parametrize("center", [False, True])
def test_maps_and_new_matrix_are_consistent(center):
    """alpha = 0: every pixel of the undistorted image has a source inside the distorted image (no black border), and a
    map entry is the forward model of its pixel: undistorting the mapped position gives the pixel back."""
    Kn = ud.get_optimal_new_camera_matrix(K, TUM_FR1, (W, H), 0, (W, H), center_principal_point=center)
    mapx, mapy = ud.init_undistort_rectify_map(K, TUM_FR1, Kn, (W, H))
    assert mapx.dtype == np.float32 and mapx.shape == (H, W)
    assert mapx.min() > -1.0 and mapx.max() < W and mapy.min() > -1.0 and mapy.max() < H
    if center:
        assert Kn[0, 2] == (W - 1) / 2 and Kn[1, 2] == (H - 1) / 2
    sel = (slice(None, None, 37), slice(None, None, 41))
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    src = np.stack((mapx[sel].ravel(), mapy[sel].ravel()), 1).astype(np.float64)
    back = ud.undistort_points(src, K, TUM_FR1, P=Kn, iters=30)
    np.testing.assert_allclose(back, np.stack((jj[sel].ravel(), ii[sel].ravel()), 1), atol=2e-3)


def test_remap_reference_interpolates_and_pads():
    img = np.zeros((4, 5), np.uint8)
    img[1, 2] = 200
    mx = np.array([[2.0, 2.5, 1.75, -3.0, 4.5]], np.float32)
    my = np.array([[1.0, 1.0, 1.5, 1.0, 1.0]], np.float32)
    out = ud.remap_reference(img, mx, my)
    # exact tap, half way to a zero neighbour, (0.75, 0.5) weights = 200 * 0.75 * 0.5, far outside, right edge (tap beyond = 0)
    assert out.tolist() == [[200, 100, 75, 0, 0]]
