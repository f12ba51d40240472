"""Lens undistortion for the calibrated dataset readers (mast3r_slam/dataloader.py:476-516: Intrinsics.from_calib calls
cv2.getOptimalNewCameraMatrix + cv2.initUndistortRectifyMap once per dataset and cv2.remap once per image).

OpenCV is not installed here, so the three functions follow OpenCV's PUBLISHED algorithms (modules/calib3d/src/
calibration.cpp: getOptimalNewCameraMatrix / icvGetRectangles; modules/calib3d/src/undistort.dispatch.cpp:
undistortPoints, initUndistortRectifyMap; modules/imgproc/src/imgwarp.cpp: remap).  **Parity with the library is unpinned**
(nothing here can run it); what the tests pin are properties: zero distortion gives K back and identity maps, distorting a
pixel grid with the forward model and undistorting it returns it to < 0.01 px, the remap kernel equals a NumPy
restatement of the same fixed-point arithmetic bit for bit.

The map construction is a one-off per dataset (float64 NumPy on the host, as OpenCV does it on the host); the per-image
`remap` runs on the device (csrc/undistort.hip).  Distortion coefficients in OpenCV's order (k1, k2, p1, p2[, k3[, k4, k5,
k6[, s1, s2, s3, s4]]])."""
import numpy as np


def _coeffs(dist):
    k = np.zeros(12)
    d = np.asarray(dist, np.float64).reshape(-1)
    k[:min(len(d), 12)] = d[:12]
    return k


def distort_points(xy, dist):
    """Forward model on normalised coordinates (n, 2) -> distorted normalised coordinates (the inner part of
    initUndistortRectifyMap / projectPoints)."""
    k = _coeffs(dist)
    x, y = xy[:, 0], xy[:, 1]
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2
    yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2
    return np.stack((xd, yd), 1)


def undistort_points(uv, K, dist, P=None, iters=5):
    """cv2.undistortPoints for pixel points (n, 2): normalise with K, invert the distortion model by `iters` fixed-point
    iterations (OpenCV's default criteria: 5), re-project with P (3x3) if given."""
    k = _coeffs(dist)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0, y0 = (uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    if P is not None:
        x, y = x * P[0, 0] + P[0, 2], y * P[1, 1] + P[1, 2]
    return np.stack((x, y), 1)


def _rectangles(K, dist, new_K, W, H, N=9):
    """icvGetRectangles: the N x N pixel grid undistorted; inner = the largest axis-aligned rectangle inside the image of
    the border points, outer = their bounding box.  -> (inner (x, y, w, h), outer (x, y, w, h))."""
    jj, ii = np.meshgrid(np.arange(N), np.arange(N))
    pts = np.stack((jj.ravel() * (W - 1) / (N - 1), ii.ravel() * (H - 1) / (N - 1)), 1).astype(np.float64)
    und = undistort_points(pts, K, dist, new_K).reshape(N, N, 2)
    x, y = und[..., 0], und[..., 1]
    ix0, ix1 = x[:, 0].max(), x[:, N - 1].min()
    iy0, iy1 = y[0, :].max(), y[N - 1, :].min()
    inner = (ix0, iy0, ix1 - ix0, iy1 - iy0)
    outer = (x.min(), y.min(), x.max() - x.min(), y.max() - y.min())
    return inner, outer


def get_optimal_new_camera_matrix(K, dist, size, alpha=0.0, new_size=None, center_principal_point=False):
    """cv2.getOptimalNewCameraMatrix(K, dist, (W, H), alpha, (W, H), centerPrincipalPoint) -> new 3x3 matrix."""
    K = np.asarray(K, np.float64)
    W, H = size
    nW, nH = new_size if new_size is not None else size
    M = K.copy()
    if center_principal_point:
        cx0, cy0 = M[0, 2], M[1, 2]
        cx, cy = (nW - 1) * 0.5, (nH - 1) * 0.5
        inner, outer = _rectangles(K, dist, K, W, H)
        s0 = max(max(cx / (cx0 - inner[0]), cy / (cy0 - inner[1])),
                 max(cx / (inner[0] + inner[2] - cx0), cy / (inner[1] + inner[3] - cy0)))
        s1 = min(min(cx / (cx0 - outer[0]), cy / (cy0 - outer[1])),
                 min(cx / (outer[0] + outer[2] - cx0), cy / (outer[1] + outer[3] - cy0)))
        s = s0 * (1 - alpha) + s1 * alpha
        M[0, 0] *= s
        M[1, 1] *= s
        M[0, 2], M[1, 2] = cx, cy
        return M
    inner, outer = _rectangles(K, dist, None, W, H)
    fx0, fy0 = (nW - 1) / inner[2], (nH - 1) / inner[3]
    cx0, cy0 = -fx0 * inner[0], -fy0 * inner[1]
    fx1, fy1 = (nW - 1) / outer[2], (nH - 1) / outer[3]
    cx1, cy1 = -fx1 * outer[0], -fy1 * outer[1]
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    return M


def init_undistort_rectify_map(K, dist, new_K, size):
    """cv2.initUndistortRectifyMap(K, dist, None, new_K, (W, H), CV_32FC1) -> (mapx, mapy) float32 (H, W): for every pixel of
    the undistorted image the position in the distorted one (float64 arithmetic, rounded once)."""
    K, new_K = np.asarray(K, np.float64), np.asarray(new_K, np.float64)
    W, H = size
    ir = np.linalg.inv(new_K)
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    _x = ir[0, 0] * jj + ir[0, 1] * ii + ir[0, 2]
    _y = ir[1, 0] * jj + ir[1, 1] * ii + ir[1, 2]
    _w = ir[2, 0] * jj + ir[2, 1] * ii + ir[2, 2]
    xy = np.stack(((_x / _w).ravel(), (_y / _w).ravel()), 1)
    d = distort_points(xy, dist)
    mapx = (K[0, 0] * d[:, 0] + K[0, 2]).reshape(H, W).astype(np.float32)
    mapy = (K[1, 1] * d[:, 1] + K[1, 2]).reshape(H, W).astype(np.float32)
    return mapx, mapy


def remap_reference(img, mapx, mapy):
    """NumPy restatement of the kernel's integer arithmetic (OpenCV's remapBilinear for 8-bit images, constant border 0):
    the checker of the device kernel, and the fallback-free host form for tests."""
    img = np.asarray(img)
    H, W = img.shape[:2]
    src = img.reshape(H, W, -1).astype(np.int64)
    sx = np.rint(mapx.astype(np.float32) * np.float32(32.0)).astype(np.int64)
    sy = np.rint(mapy.astype(np.float32) * np.float32(32.0)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.zeros(mapx.shape + (src.shape[2],), np.int64)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        yy, xx = iy + dy, ix + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)] * ok[..., None]
        acc += v * wk[..., None]
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out.reshape(mapx.shape + img.shape[2:])


class DeviceRemap:
    """cv2.remap(img, mapx, mapy, INTER_LINEAR) for 8-bit images on the device; the maps are uploaded once."""

    def __init__(self, mapx, mapy, device="cuda"):
        import torch

        self.device = torch.device(device)
        self.shape = mapx.shape
        self.mapx = torch.from_numpy(np.ascontiguousarray(mapx, np.float32)).to(self.device)
        self.mapy = torch.from_numpy(np.ascontiguousarray(mapy, np.float32)).to(self.device)

    def __call__(self, img):
        import torch

        import mslam_hip as _m

        img = np.ascontiguousarray(img)
        if img.dtype != np.uint8:
            raise TypeError("remap: 8-bit images only (the dataset readers decode to uint8)")
        H, W = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        src = torch.from_numpy(img).to(self.device)
        dst = torch.empty(self.shape + ((ch,) if img.ndim == 3 else ()), dtype=torch.uint8, device=self.device)
        rc = _m.lib().mslam_remap_bilinear_u8(_m.ptr(src), H, W, ch, _m.ptr(self.mapx), _m.ptr(self.mapy), _m.ptr(dst),
                                              self.shape[0], self.shape[1], _m.stream_ptr())
        _m.check(rc, "remap_bilinear_u8")
        return dst.cpu().numpy()
