"""Compute part of mast3r_slam/tsdf/global_manager.py: TSDFGlobalIntegrator._integrate_single (lines 81-106)
and the after-solve hook.  The daemon threads, dirty queue and logging of the reference classes
(lines 17-79, 116-190) are scheduling and stay out of scope; callers invoke these synchronously (bench.py runs
them on its backend thread)."""
import torch

from lietorch_hip import Sim3


class TSDFGlobalIntegrator:
    def __init__(self, volume, optimizer, keyframes, cfg):
        self.volume = volume
        self.optimizer = optimizer
        self.keyframes = keyframes
        self.cfg = cfg
        self.max_points = int(cfg.get("max_points_per_kf", 40000))
        self.min_conf = float(cfg.get("min_confidence", 0.05))
        self.next_idx = 0

    def _integrate_single(self, idx):
        """global_manager.py:81-106: random subset (<= max_points_per_kf) of the points with C > min_confidence,
        moved to the world frame with T_WC, fused with the camera centre as ray origin.  Everything stays on
        the device (the reference round-trips through numpy because its volume is a python dict)."""
        if idx >= len(self.keyframes):
            return
        frame = self.keyframes[idx]
        points = frame.X_canon.detach().reshape(-1, 3)
        conf = frame.C.detach().reshape(-1)
        valid_idx = torch.nonzero(conf > self.min_conf).view(-1)
        if valid_idx.numel() == 0:
            return
        count = min(valid_idx.numel(), self.max_points)
        choice = valid_idx[torch.randperm(valid_idx.numel(), device=valid_idx.device)[:count]]
        pose = Sim3(frame.T_WC.data.clone())
        pts_world = pose.act(points[choice].contiguous())
        cam_origin = pose.act(torch.zeros(1, 3, device=points.device, dtype=points.dtype)).squeeze(0)
        self.volume.integrate(pts_world, conf[choice].double(), cam_origin, return_fused=False)

    def _integrate_new_keyframes(self):
        """global_manager.py:64-69."""
        while self.next_idx < len(self.keyframes):
            self._integrate_single(self.next_idx)
            self.optimizer.pre_refine(self.next_idx)
            self.next_idx += 1
