"""Mirror of the hot-path part of mast3r_slam/frame.py: the Frame record (lines 17-108, the data-layout
contract of every per-keyframe tensor) and create_frame (lines 111-122).  The CUDA-IPC shared ring
buffers (SharedStates / SharedKeyframes, lines 125-334) belong to the reference's 3-process runtime and
are out of scope; `KeyframeStore` below is the single-process container the hot path and
the bench use, with the same field names and shapes and no 110-slot cap."""
import dataclasses
from enum import Enum
from typing import Optional

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

    def __len__(self):
        return len(self._kfs)

    def __getitem__(self, idx):
        return self._kfs[int(idx)]

    def __setitem__(self, idx, frame):
        self._kfs[int(idx)] = frame

    def append(self, frame):
        self._kfs.append(frame)

    def last_keyframe(self):
        return self._kfs[-1] if self._kfs else None

    def update_T_WCs(self, T_WCs, idx):
        for k, i in enumerate(idx.tolist() if hasattr(idx, "tolist") else idx):
            self._kfs[int(i)].T_WC = Sim3(T_WCs.data[k].reshape(1, 8).clone())
