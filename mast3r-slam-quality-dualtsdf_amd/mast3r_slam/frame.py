"""Mirror of the hot-path part of mast3r_slam/frame.py: the Frame record (lines 17-108, the data-layout
contract of every per-keyframe tensor), create_frame (lines 111-122) and the two buffer classes
SharedStates / SharedKeyframes (lines 125-334) with the same fields, shapes, dtypes and method semantics.
The reference shares those buffers between three processes (torch.multiprocessing manager + CUDA IPC); here
one process drives the GPU from host threads (bench.py), so `manager` may be None (threading.RLock, plain
values) and the slot count is a constructor argument sized for 288 GB of HBM instead of the 110-slot cap.
`KeyframeStore` is the list-like container the unit tests and the bench use."""
import dataclasses
import threading
from enum import Enum
from typing import Optional

import copy

import torch

from lietorch_hip import Sim3
from mast3r_slam.config import config


class Mode(Enum):
    INIT = 0
    TRACKING = 1
    RELOC = 2
    TERMINATED = 3


@dataclasses.dataclass
class Frame:
    frame_id: int
    img: torch.Tensor                 # (1,3,H,W) f32, ImgNorm range
    img_shape: torch.Tensor           # (1,2) i32
    img_true_shape: torch.Tensor      # (1,2) i32
    uimg: Optional[torch.Tensor]      # (H,W,3) f32 CPU
    T_WC: Sim3 = None                 # (1,8) [t, q(xyzw), s]
    X_canon: Optional[torch.Tensor] = None   # (HW,3)
    C: Optional[torch.Tensor] = None         # (HW,1) running SUM of confidences (mean = C / N)
    feat: Optional[torch.Tensor] = None      # (1,N,1024)
    pos: Optional[torch.Tensor] = None       # (1,N,2) i64
    N: int = 0
    N_updates: int = 0
    K: Optional[torch.Tensor] = None

    def __post_init__(self):
        if self.T_WC is None:
            self.T_WC = Sim3.Identity(1, device=self.img.device)

    def get_score(self, C):
        return torch.median(C) if config["tracking"]["filtering_score"] == "median" else torch.mean(C)

    def update_pointmap(self, X, C):
        """frame.py:41-105 (all filtering modes)."""
        mode = config["tracking"]["filtering_mode"]
        if self.N == 0:
            self.X_canon, self.C, self.N, self.N_updates = X.clone(), C.clone(), 1, 1
            if mode == "best_score":
                self.score = self.get_score(C)
            return
        if mode == "first":
            if self.N_updates == 1:
                self.X_canon, self.C, self.N = X.clone(), C.clone(), 1
        elif mode == "recent":
            self.X_canon, self.C, self.N = X.clone(), C.clone(), 1
        elif mode == "best_score":
            new_score = self.get_score(C)
            if new_score > self.score:
                self.X_canon, self.C, self.N, self.score = X.clone(), C.clone(), 1, new_score
        elif mode == "indep_conf":
            new_mask = C > self.C
            self.X_canon[new_mask.repeat(1, 3)] = X[new_mask.repeat(1, 3)]
            self.C[new_mask] = C[new_mask]
            self.N = 1
        elif mode == "weighted_pointmap":
            self.X_canon = ((self.C * self.X_canon) + (C * X)) / (self.C + C)
            self.C = self.C + C
            self.N += 1
        elif mode == "weighted_spherical":
            def to_sph(P):
                r = torch.linalg.norm(P, dim=-1, keepdim=True)
                x, y, z = torch.tensor_split(P, 3, dim=-1)
                return torch.cat((r, torch.atan2(y, x), torch.acos(z / r)), dim=-1)

            def to_cart(s):
                r, phi, theta = torch.tensor_split(s, 3, dim=-1)
                return torch.cat((r * torch.sin(theta) * torch.cos(phi), r * torch.sin(theta) * torch.sin(phi),
                                  r * torch.cos(theta)), dim=-1)

            sph = ((self.C * to_sph(self.X_canon)) + (C * to_sph(X))) / (self.C + C)
            self.X_canon = to_cart(sph)
            self.C = self.C + C
            self.N += 1
        self.N_updates += 1

    def get_average_conf(self):
        return self.C / self.N if self.C is not None else None


def create_frame(i, img, T_WC, img_size=512, device="cuda:0"):
    """frame.py:111-122."""
    from mast3r_slam.mast3r_utils import resize_img

    res = resize_img(img, img_size)
    rgb = res["img"].to(device=device)
    img_shape = torch.tensor(res["true_shape"], device=device)
    img_true_shape = img_shape.clone()
    uimg = torch.from_numpy(res["unnormalized_img"]) / 255.0
    ds = config["dataset"]["img_downsample"]
    if ds > 1:
        uimg = uimg[::ds, ::ds]
        img_shape = img_shape // ds
    return Frame(i, rgb, img_shape, img_true_shape, uimg, T_WC)


class KeyframeStore:
    """Single-process keyframe container (list semantics of SharedKeyframes: append, [], len,
    last_keyframe, update_T_WCs; frame.py:254-334) without the IPC buffers or the 110-slot cap."""

    def __init__(self):
        self._kfs = []
        self._stamps = []      # per keyframe: bumped whenever its pointmap / confidence may have changed
        self._clock = 0
        self.K = None          # calibrated runs: SharedKeyframes.set_intrinsics (frame.py:325-333)

    def set_intrinsics(self, K):
        assert config["use_calib"]
        self.K = K
        for kf in self._kfs:
            kf.K = K

    def get_intrinsics(self):
        assert config["use_calib"]
        return self.K

    def __len__(self):
        return len(self._kfs)

    def __getitem__(self, idx):
        return self._kfs[int(idx)]

    def __setitem__(self, idx, frame):
        self._kfs[int(idx)] = frame
        self.touch(idx)

    def append(self, frame):
        """Copy-in, as SharedKeyframes does (frame.py:312-314): later updates of the keyframe (pointmap fusion, backend
        poses) do not reach back into the caller's Frame object.  Tensors are shared, not cloned."""
        self._kfs.append(copy.copy(frame))
        if self.K is not None and config["use_calib"]:
            self._kfs[-1].K = self.K        # a keyframe read from the store carries the intrinsics (frame.py:262-263)
        self._stamps.append(0)
        self.touch(len(self._kfs) - 1)

    def pop_last(self):
        self._kfs.pop()
        self._stamps.pop()

    def touch(self, idx):
        """Note that keyframe `idx`'s pointmap / confidence changed (assignments do it themselves; code that edits the
        tensors in place - the local TSDF refiner - calls it).  A sharded backend re-sends touched keyframes."""
        self._clock += 1
        self._stamps[int(idx)] = self._clock

    def stamp(self, idx):
        return self._stamps[int(idx)]

    def last_keyframe(self):
        return self._kfs[-1] if self._kfs else None

    def update_T_WCs(self, T_WCs, idx):
        data = T_WCs.data.reshape(-1, 8).clone()      # one copy; every keyframe gets a view of it (poses are never
        for k, i in enumerate(idx.tolist() if hasattr(idx, "tolist") else idx):   # written in place)
            self._kfs[int(i)].T_WC = Sim3(data[k].reshape(1, 8))


class _Value:
    """manager.Value stand-in for single-process use."""

    def __init__(self, value=0):
        self.value = value


def _lock(manager):
    return manager.RLock() if manager is not None else threading.RLock()


def _value(manager, v):
    return manager.Value("i", v) if manager is not None else _Value(v)


def _list(manager):
    return manager.list() if manager is not None else []


class SharedStates:
    """frame.py:125-217: the current frame (for relocalisation / visualisation) plus the mode, pause flag,
    relocalisation semaphore, backend task list and edge lists."""

    def __init__(self, manager, h, w, dtype=torch.float32, device="cuda"):
        self.h, self.w = h, w
        self.dtype = dtype
        self.device = device
        self.lock = _lock(manager)
        self.paused = _value(manager, 0)
        self.mode = _value(manager, Mode.INIT)
        self.reloc_sem = _value(manager, 0)
        self.global_optimizer_tasks = _list(manager)
        self.edges_ii = _list(manager)
        self.edges_jj = _list(manager)
        self.feat_dim = 1024
        self.num_patches = h * w // (16 * 16)
        self.dataset_idx = torch.zeros(1, device=device, dtype=torch.int)
        self.img = torch.zeros(3, h, w, device=device, dtype=dtype)
        self.uimg = torch.zeros(h, w, 3, device="cpu", dtype=dtype)
        self.img_shape = torch.zeros(1, 2, device=device, dtype=torch.int)
        self.img_true_shape = torch.zeros(1, 2, device=device, dtype=torch.int)
        self.T_WC = Sim3.Identity(1, device=device).data.to(dtype)
        self.X = torch.zeros(h * w, 3, device=device, dtype=dtype)
        self.C = torch.zeros(h * w, 1, device=device, dtype=dtype)
        self.feat = torch.zeros(1, self.num_patches, self.feat_dim, device=device, dtype=dtype)
        self.pos = torch.zeros(1, self.num_patches, 2, device=device, dtype=torch.long)

    def set_frame(self, frame):
        with self.lock:
            self.dataset_idx[:] = frame.frame_id
            self.img[:] = frame.img
            self.uimg[:] = frame.uimg
            self.img_shape[:] = frame.img_shape
            self.img_true_shape[:] = frame.img_true_shape
            self.T_WC[:] = frame.T_WC.data
            self.X[:] = frame.X_canon
            self.C[:] = frame.C
            self.feat[:] = frame.feat
            self.pos[:] = frame.pos

    def get_frame(self):
        with self.lock:
            frame = Frame(int(self.dataset_idx[0]), self.img, self.img_shape, self.img_true_shape, self.uimg,
                          Sim3(self.T_WC))
            frame.X_canon = self.X
            frame.C = self.C
            frame.feat = self.feat
            frame.pos = self.pos
            return frame

    def queue_global_optimization(self, idx):
        with self.lock:
            self.global_optimizer_tasks.append(idx)

    def queue_reloc(self):
        with self.lock:
            self.reloc_sem.value += 1

    def dequeue_reloc(self):
        with self.lock:
            if self.reloc_sem.value == 0:
                return
            self.reloc_sem.value -= 1

    def get_mode(self):
        with self.lock:
            return self.mode.value

    def set_mode(self, mode):
        with self.lock:
            self.mode.value = mode

    def pause(self):
        with self.lock:
            self.paused.value = 1

    def unpause(self):
        with self.lock:
            self.paused.value = 0

    def is_paused(self):
        with self.lock:
            return self.paused.value == 1


class SharedKeyframes:
    """frame.py:220-334: slot buffers for every per-keyframe tensor; `kf = keyframes[i]` returns a Frame of VIEWS
    into the buffers, `keyframes[i] = frame` copies a frame in and marks the slot dirty.  One keyframe is
    ~9.4 MB of HBM at 512x384 (img 2.4, X 2.4, C 0.8, feat 3.1, pos 0.01 + 2.4 MB of host memory for uimg), so
    the reference's 110-slot cap (1 GB) is only a default here: 10 000-frame runs pass buffer=1250+ (12 GB)."""

    def __init__(self, manager, h, w, buffer=110, dtype=torch.float32, device="cuda"):
        self.lock = _lock(manager)
        self.n_size = _value(manager, 0)
        self.h, self.w = h, w
        self.buffer = buffer
        self.dtype = dtype
        self.device = device
        self.feat_dim = 1024
        self.num_patches = h * w // (16 * 16)
        self.frame_id_to_index = {}
        self.dataset_idx = torch.zeros(buffer, device=device, dtype=torch.int)
        self.img = torch.zeros(buffer, 3, h, w, device=device, dtype=dtype)
        self.uimg = torch.zeros(buffer, h, w, 3, device="cpu", dtype=dtype)
        self.img_shape = torch.zeros(buffer, 1, 2, device=device, dtype=torch.int)
        self.img_true_shape = torch.zeros(buffer, 1, 2, device=device, dtype=torch.int)
        self.T_WC = torch.zeros(buffer, 1, Sim3.embedded_dim, device=device, dtype=dtype)
        self.X = torch.zeros(buffer, h * w, 3, device=device, dtype=dtype)
        self.C = torch.zeros(buffer, h * w, 1, device=device, dtype=dtype)
        self.N = torch.zeros(buffer, device=device, dtype=torch.int)
        self.N_updates = torch.zeros(buffer, device=device, dtype=torch.int)
        self.feat = torch.zeros(buffer, 1, self.num_patches, self.feat_dim, device=device, dtype=dtype)
        self.pos = torch.zeros(buffer, 1, self.num_patches, 2, device=device, dtype=torch.long)
        self.is_dirty = torch.zeros(buffer, 1, device=device, dtype=torch.bool)
        self.K = torch.zeros(3, 3, device=device, dtype=dtype)
        self.version = torch.zeros(buffer, device=device, dtype=torch.long)
        self._stamps, self._clock = {}, 0      # host-side change stamps (KeyframeStore.touch / stamp)

    def touch(self, idx):
        self._clock += 1
        self._stamps[int(idx)] = self._clock

    def stamp(self, idx):
        return self._stamps.get(int(idx))

    def __getitem__(self, idx) -> Frame:
        with self.lock:
            kf = Frame(int(self.dataset_idx[idx]), self.img[idx], self.img_shape[idx], self.img_true_shape[idx],
                       self.uimg[idx], Sim3(self.T_WC[idx]))
            kf.X_canon = self.X[idx]
            kf.C = self.C[idx]
            kf.feat = self.feat[idx]
            kf.pos = self.pos[idx]
            kf.N = int(self.N[idx])
            kf.N_updates = int(self.N_updates[idx])
            if config["use_calib"]:
                kf.K = self.K
            return kf

    def __setitem__(self, idx, value: Frame) -> None:
        with self.lock:
            if idx >= self.buffer:
                raise IndexError(f"SharedKeyframes: slot {idx} exceeds the buffer of {self.buffer} keyframes "
                                 "(pass a larger `buffer`; ~9.4 MB of HBM per slot at 512x384)")
            self.n_size.value = max(idx + 1, self.n_size.value)
            self.frame_id_to_index[value.frame_id] = idx
            self.dataset_idx[idx] = value.frame_id
            self.img[idx] = value.img
            self.uimg[idx] = value.uimg
            self.img_shape[idx] = value.img_shape
            self.img_true_shape[idx] = value.img_true_shape
            self.T_WC[idx] = value.T_WC.data
            self.X[idx] = value.X_canon
            self.C[idx] = value.C
            self.feat[idx] = value.feat
            self.pos[idx] = value.pos
            self.N[idx] = value.N
            self.N_updates[idx] = value.N_updates
            self.is_dirty[idx] = True
            self.touch(idx)
            return idx

    def __len__(self):
        with self.lock:
            return self.n_size.value

    def append(self, value: Frame):
        with self.lock:
            self[self.n_size.value] = value

    def pop_last(self):
        with self.lock:
            self.n_size.value -= 1

    def last_keyframe(self) -> Optional[Frame]:
        with self.lock:
            if self.n_size.value == 0:
                return None
            return self[self.n_size.value - 1]

    def update_T_WCs(self, T_WCs, idx) -> None:
        with self.lock:
            self.T_WC[idx] = T_WCs.data

    def get_dirty_idx(self):
        with self.lock:
            idx = torch.where(self.is_dirty)[0]
            self.is_dirty[:] = False
            return idx

    def set_intrinsics(self, K):
        assert config["use_calib"]
        with self.lock:
            self.K[:] = K

    def get_intrinsics(self):
        assert config["use_calib"]
        with self.lock:
            return self.K
