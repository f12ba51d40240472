"""Mirror of the file-based readers of the reference's mast3r_slam/dataloader.py: same class names,
attributes (`rgb_files`, `timestamps`, `img_size`, `camera_intrinsics`, `use_calibration`, `save_results`,
`dataset_path`) and `dataset[i] -> (timestamp, HxWx3 float32 RGB in [0,1])`, so `main.py`'s loop and `evaluate.py`'s
writers run unchanged on top of it.  Images are decoded with PIL (the reference uses cv2.imread + BGR->RGB: the same RGB
bytes for PNG; JPEG decoders may differ in the last bit).

What is NOT here, and fails loudly instead of silently doing something else:
* lens undistortion (`Intrinsics.from_calib` -> cv2.getOptimalNewCameraMatrix / initUndistortRectifyMap / remap,
  dataloader.py:497-516): OpenCV is not installed; since round 3 the three functions are rebuilt from OpenCV's published
  algorithms (mast3r_slam/undistort.py; the per-image remap, 5-bit fixed-point weights included, as a HIP kernel), so
  calibrated datasets (TUM / EuRoC / ETH3D / 7-Scenes) load - parity with cv2 itself is UNPINNED, property-tested only.
* live sources (RealsenseDataset, Webcam) and MP4Dataset (pyrealsense2 / cv2.VideoCapture / torchcodec).
`SyntheticRoomDataset` is an addition: the procedural room of mast3r_slam.synthetic as a dataset (BASELINE configs 3, 5)."""
import json
import pathlib
import re

import numpy as np

from mast3r_slam.config import config
from mast3r_slam.mast3r_utils import resize_img


def natsorted(paths):
    """Digit-aware ordering of path names (the reference uses natsort.natsorted): frame-2 before frame-10."""
    key = lambda p: [int(t) if t.isdigit() else t.lower() for t in re.split(r"(\d+)", str(p))]
    return sorted(paths, key=key)


class MonocularDataset:
    """dataloader.py:22-66."""

    def __init__(self, dtype=np.float32):
        self.dtype = dtype
        self.rgb_files = []
        self.timestamps = []
        self.img_size = 512
        self.camera_intrinsics = None
        self.use_calibration = config["use_calib"]
        self.save_results = True

    def __len__(self):
        return len(self.rgb_files)

    def __getitem__(self, idx):
        img = self.get_image(idx)           # image first: live sources create the timestamp while reading
        return self.get_timestamp(idx), img

    def get_timestamp(self, idx):
        return self.timestamps[idx]

    def read_img(self, idx):
        import PIL.Image

        return np.asarray(PIL.Image.open(str(self.rgb_files[idx])).convert("RGB"))

    def get_image(self, idx):
        img = self.read_img(idx)
        if self.use_calibration:
            img = self.camera_intrinsics.remap(img)
        return img.astype(self.dtype) / 255.0

    def get_img_shape(self):
        img = self.read_img(0)
        raw_img_shape = img.shape
        img = resize_img(img, self.img_size)
        return img["img"][0].shape[1:], raw_img_shape[:2]

    def subsample(self, subsample):
        self.rgb_files = self.rgb_files[::subsample]
        self.timestamps = self.timestamps[::subsample]

    def has_calib(self):
        return self.camera_intrinsics is not None


def _read_list(path, delimiter):
    """`timestamp<delimiter>file` lines as a 2-column string array (np.loadtxt(dtype=str) in the reference: '#' comments
    and blank lines are skipped)."""
    return np.loadtxt(path, delimiter=delimiter, dtype=str, skiprows=0, ndmin=2)


class TUMDataset(MonocularDataset):
    """dataloader.py:69-91: rgb.txt, calibration by freiburg camera number."""

    CALIB = {1: [517.3, 516.5, 318.6, 255.3, 0.2624, -0.9531, -0.0054, 0.0026, 1.1633],
             2: [520.9, 521.0, 325.1, 249.7, 0.2312, -0.7849, -0.0033, -0.0001, 0.9172],
             3: [535.4, 539.2, 320.1, 247.6]}

    def __init__(self, dataset_path):
        super().__init__()
        self.dataset_path = pathlib.Path(dataset_path)
        tstamp_rgb = _read_list(self.dataset_path / "rgb.txt", " ")
        self.rgb_files = [self.dataset_path / f for f in tstamp_rgb[:, 1]]
        self.timestamps = tstamp_rgb[:, 0]
        idx = int(re.search(r"freiburg(\d+)", str(dataset_path)).group(1))
        self.camera_intrinsics = Intrinsics.from_calib(self.img_size, 640, 480, np.array(self.CALIB[idx]))


class EurocDataset(MonocularDataset):
    """dataloader.py:94-118: always undistorted in the reference, hence unusable without OpenCV's maps."""

    def __init__(self, dataset_path):
        super().__init__()
        self.use_calibration = True
        self.dataset_path = pathlib.Path(dataset_path)
        tstamp_rgb = _read_list(self.dataset_path / "mav0/cam0/data.csv", ",")
        self.rgb_files = [self.dataset_path / "mav0/cam0/data" / f for f in tstamp_rgb[:, 1]]
        self.timestamps = tstamp_rgb[:, 0]
        import yaml

        with open(self.dataset_path / "mav0/cam0/sensor.yaml") as f:
            self.cam0 = yaml.load(f, Loader=yaml.SafeLoader)
        W, H = self.cam0["resolution"]
        self.camera_intrinsics = Intrinsics.from_calib(
            self.img_size, W, H, [*self.cam0["intrinsics"], *self.cam0["distortion_coefficients"]], always_undistort=True)

    def read_img(self, idx):
        import PIL.Image

        return np.asarray(PIL.Image.open(str(self.rgb_files[idx])).convert("L").convert("RGB"))


class ETH3DDataset(MonocularDataset):
    """dataloader.py:121-136."""

    def __init__(self, dataset_path):
        super().__init__()
        self.dataset_path = pathlib.Path(dataset_path)
        tstamp_rgb = _read_list(self.dataset_path / "rgb.txt", " ")
        self.rgb_files = [self.dataset_path / f for f in tstamp_rgb[:, 1]]
        self.timestamps = tstamp_rgb[:, 0]
        calibration = np.loadtxt(self.dataset_path / "calibration.txt", delimiter=" ", dtype=np.float32, skiprows=0)
        _, (H, W) = self.get_img_shape()
        self.camera_intrinsics = Intrinsics.from_calib(self.img_size, W, H, calibration)


class SevenScenesDataset(MonocularDataset):
    """dataloader.py:139-150."""

    def __init__(self, dataset_path):
        super().__init__()
        self.dataset_path = pathlib.Path(dataset_path)
        self.rgb_files = natsorted(list((self.dataset_path / "seq-01").glob("*.color.png")))
        self.timestamps = np.arange(0, len(self.rgb_files)).astype(self.dtype)
        self.camera_intrinsics = Intrinsics.from_calib(self.img_size, 640, 480, [585.0, 585.0, 320.0, 240.0])


class RGBFiles(MonocularDataset):
    """dataloader.py:267-273."""

    def __init__(self, dataset_path):
        super().__init__()
        self.use_calibration = False
        self.dataset_path = pathlib.Path(dataset_path)
        self.rgb_files = natsorted(list(self.dataset_path.glob("*.png")))
        self.timestamps = np.arange(0, len(self.rgb_files)).astype(self.dtype) / 30.0


class ReplicaDataset(MonocularDataset):
    """dataloader.py:276-474: <seq>/results/frame*.jpg|png (depth*.png ignored), timestamps from the first column of
    traj.txt when it has enough rows (else 30 fps), intrinsics from a cam_params.json next to / above the sequence."""

    def __init__(self, dataset_path):
        super().__init__()
        self.dataset_path = pathlib.Path(dataset_path)
        img_dir = self.dataset_path / "results"
        if not img_dir.is_dir():
            img_dir = self.dataset_path
        dirs = [img_dir / sub for sub in ("color", "rgb") if (img_dir / sub).is_dir()] + [img_dir]
        patterns = [f"frame*.{ext}" for ext in ("jpg", "JPG", "jpeg", "JPEG", "png", "PNG")]
        found = []
        for d in dirs:
            found = [p for pat in patterns for p in d.glob(pat)]
            if found:
                break
        if not found:
            found = [p for d in dirs for pat in patterns for p in d.rglob(pat)]
        found = {p for p in found if "depth" not in p.name.lower()}
        self.rgb_files = [str(p) for p in natsorted(found)]
        if not self.rgb_files:
            raise FileNotFoundError(f"[ReplicaDataset] No RGB frames (frame*.jpg/png) found under {img_dir} (and subdirs).")
        H, W = self.read_img(0).shape[:2]
        ts = None
        traj = self.dataset_path / "traj.txt"
        if traj.is_file():
            try:
                arr = np.loadtxt(str(traj), dtype=np.float64)
                arr = arr.reshape(-1, 1) if arr.ndim == 1 else arr
                if arr.shape[0] >= len(self.rgb_files):
                    ts = arr[: len(self.rgb_files), 0].astype(self.dtype)
            except Exception:
                ts = None
        self.timestamps = np.arange(len(self.rgb_files), dtype=self.dtype) / 30.0 if ts is None else ts
        fx = fy = cx = cy = dist = None
        for p in (self.dataset_path, self.dataset_path.parent, self.dataset_path.parent.parent):
            if (p / "cam_params.json").exists():
                try:
                    cam = json.load(open(p / "cam_params.json", "r", encoding="utf-8"))
                    fx, fy, cx, cy, dist = self._parse_cam_dict(cam)
                    if fx is None and isinstance(cam, dict):
                        inner = cam.get("camera") or cam.get("rgb") or cam.get("color")
                        if isinstance(inner, dict):
                            fx, fy, cx, cy, dist = self._parse_cam_dict(inner)
                except Exception:
                    pass
                break
        if None not in (fx, fy, cx, cy):
            calib = [float(fx), float(fy), float(cx), float(cy)] + ([float(v) for v in dist] if dist else [])
        else:
            calib = [0.9 * W, 0.9 * W, W / 2.0, H / 2.0]
        self.camera_intrinsics = Intrinsics.from_calib(self.img_size, W, H, calib)

    @staticmethod
    def _parse_cam_dict(d):
        """fx/fy/cx/cy keys (+ distortion list or k1,k2,p1,p2,k3), an `intrinsics` 4-vector, or a 3x3 / flat-9 `K`."""
        none = (None, None, None, None, None)
        if not isinstance(d, dict):
            return none
        if all(k in d for k in ("fx", "fy", "cx", "cy")):
            if isinstance(d.get("distortion"), (list, tuple)):
                dist = [float(v) for v in d["distortion"]]
            else:
                dist = [float(d[k]) for k in ("k1", "k2", "p1", "p2", "k3") if k in d] or None
            return float(d["fx"]), float(d["fy"]), float(d["cx"]), float(d["cy"]), dist
        intr = d.get("intrinsics")
        if isinstance(intr, (list, tuple)) and len(intr) >= 4:
            dist = [float(v) for v in d["distortion"]] if isinstance(d.get("distortion"), (list, tuple)) else None
            return float(intr[0]), float(intr[1]), float(intr[2]), float(intr[3]), dist
        K = d.get("K")
        if isinstance(K, (list, tuple)):
            if len(K) == 9:
                return float(K[0]), float(K[4]), float(K[2]), float(K[5]), None
            if len(K) == 3 and isinstance(K[0], (list, tuple)) and len(K[0]) == 3:
                return float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), None
        return none


class SyntheticRoomDataset(MonocularDataset):
    """The procedural room (mast3r_slam.synthetic) as a dataset: frame i is rendered from camera_pose(stride * i)."""

    def __init__(self, n_frames=1000, stride=1, h=384, w=512, fps=30.0):
        super().__init__()
        self.use_calibration = False
        self.dataset_path = pathlib.Path(f"synthetic_room_{n_frames}")
        self.n_frames, self.stride, self.h, self.w = n_frames, stride, h, w
        self.rgb_files = list(range(n_frames))
        self.timestamps = np.arange(n_frames).astype(self.dtype) / fps

    def read_img(self, idx):
        from mast3r_slam import synthetic

        rgb = synthetic.render_rgb(synthetic.camera_pose(self.stride * self.rgb_files[idx]), self.h, self.w)   # (3,h,w) in [-1,1]
        return np.clip((np.transpose(rgb, (1, 2, 0)) * 0.5 + 0.5) * 255.0, 0, 255).astype(np.uint8)


class Intrinsics:
    """dataloader.py:476-516.  K_frame (the intrinsics after resize_img's scale + centre crop) is plain arithmetic; the
    undistortion maps and the per-image remap follow OpenCV's published algorithms (mast3r_slam/undistort.py, the remap
    as a HIP kernel) - OpenCV itself is not installed, so parity with cv2 is UNPINNED (property-tested only)."""

    def __init__(self, img_size, W, H, K_orig, K, distortion, mapx, mapy):
        self.img_size = img_size
        self.W, self.H = W, H
        self.K_orig, self.K, self.distortion, self.mapx, self.mapy = K_orig, K, distortion, mapx, mapy
        _, (scale_w, scale_h, half_crop_w, half_crop_h) = resize_img(np.zeros((H, W, 3)), self.img_size, return_transformation=True)
        self.K_frame = self.K.copy()
        self.K_frame[0, 0] = self.K[0, 0] / scale_w
        self.K_frame[1, 1] = self.K[1, 1] / scale_h
        self.K_frame[0, 2] = self.K[0, 2] / scale_w - half_crop_w
        self.K_frame[1, 2] = self.K[1, 2] / scale_h - half_crop_h
        self._remap = None

    def remap(self, img):
        """cv2.remap(img, mapx, mapy, cv2.INTER_LINEAR) (dataloader.py:495-496) on the device."""
        if self.mapx is None:
            raise ValueError("Intrinsics.remap: no undistortion maps (constructed for already undistorted images)")
        if self._remap is None:
            from mast3r_slam.undistort import DeviceRemap

            self._remap = DeviceRemap(self.mapx, self.mapy)
        return self._remap(img)

    @staticmethod
    def from_calib(img_size, W, H, calib, always_undistort=False):
        if not config["use_calib"] and not always_undistort:
            return None
        from mast3r_slam import undistort as ud

        fx, fy, cx, cy = calib[:4]
        distortion = np.zeros(4)
        if len(calib) > 4:
            distortion = np.array(calib[4:])
        K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
        center = config["dataset"]["center_principle_point"]
        K_opt = ud.get_optimal_new_camera_matrix(K, distortion, (W, H), 0, (W, H), center_principal_point=center)
        mapx, mapy = ud.init_undistort_rectify_map(K, distortion, K_opt, (W, H))
        return Intrinsics(img_size, W, H, K, K_opt, distortion, mapx, mapy)


def load_dataset(dataset_path):
    """dataloader.py:519-548: pick the reader from the path tokens / extension."""
    tokens = [s.lower() for s in re.split(r"[\\/]+", str(dataset_path))]
    if "replica" in tokens:
        return ReplicaDataset(dataset_path)
    if "tum" in tokens:
        return TUMDataset(dataset_path)
    if "euroc" in tokens:
        return EurocDataset(dataset_path)
    if "eth3d" in tokens:
        return ETH3DDataset(dataset_path)
    if "7-scenes" in tokens or "7scenes" in tokens or "7_scenes" in tokens:
        return SevenScenesDataset(dataset_path)
    if "synthetic_room" in tokens:
        return SyntheticRoomDataset()
    if "realsense" in tokens or "webcam" in tokens:
        raise NotImplementedError("live camera sources (pyrealsense2 / cv2.VideoCapture) are not available here")
    ext = pathlib.Path(str(dataset_path)).suffix.lower()
    if ext in (".mp4", ".avi", ".mov"):
        raise NotImplementedError("video files need torchcodec or cv2.VideoCapture (not installed)")
    return RGBFiles(dataset_path)
