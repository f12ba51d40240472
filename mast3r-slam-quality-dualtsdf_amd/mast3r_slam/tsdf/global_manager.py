"""Mirror of mast3r_slam/tsdf/global_manager.py: TSDFGlobalIntegrator (lines 16-114) and TSDFGlobalManager
(lines 177-226), same constructor signatures and method names.  The reference runs the integrator and the pose
optimiser as daemon threads that poll every 0.1 s; here one synchronous pass of each runs inside
`on_after_backend_solve` on the caller's stream (bench.py / SlamSystem call it from the backend), so the order
integrate-new -> re-integrate-updated -> optimise is fixed instead of depending on thread timing."""
import torch

from lietorch_hip import Sim3
from mast3r_slam.config import config


class TSDFGlobalIntegrator:
    def __init__(self, volume, keyframes, cfg, optimizer):
        """global_manager.py:19-41."""
        self.volume = volume
        self.keyframes = keyframes
        self.cfg = cfg
        self.optimizer = optimizer
        self.max_points = int(cfg.get("max_points_per_kf", 40000))
        self.min_conf = float(cfg.get("min_confidence", 0.05))
        self.next_idx = 0
        self.max_pending = int(cfg.get("reintegration_queue", 256))
        self.pending = []     # the reference's reintegration_queue + pending set: ordered, no duplicates, bounded

    def mark_pose_update(self, indices):
        """global_manager.py:46-54."""
        for idx in indices:
            if idx in self.pending:
                continue
            if len(self.pending) >= self.max_pending:
                break
            self.pending.append(idx)

    def _process_dirty_queue(self):
        """global_manager.py:71-79: keyframes whose pose the backend moved are fused again at the new pose."""
        while self.pending:
            idx = self.pending.pop(0)
            if idx < len(self.keyframes):
                self._integrate_single(idx)

    def run_once(self):
        """One pass of the reference thread's loop body (global_manager.py:57-60)."""
        self._integrate_new_keyframes()
        self._process_dirty_queue()

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


class TSDFGlobalManager:
    """global_manager.py:177-226 -> same surface (start / shutdown / on_after_backend_solve, .volume / .optimizer /
    .integrator); no threads."""

    def __init__(self, keyframes, cfg, use_calib, device):
        from .global_volume import TSDFVolume
        from .tsdf_optimizer import TSDFPoseOptimizer

        self.enabled = bool(cfg.get("enabled", False))
        self.keyframes = keyframes
        self.cfg = cfg
        self.volume = TSDFVolume(voxel_size=cfg.get("voxel_size", 0.03), truncation=cfg.get("trunc_dist", 0.12),
                                 max_weight=cfg.get("max_weight", 100.0), min_weight=cfg.get("min_tsdf_weight", 1.0e-3),
                                 capacity=int(cfg.get("hash_capacity", 1 << 22)), device=device)
        self.optimizer = TSDFPoseOptimizer(self.volume, keyframes, cfg, use_calib, device)
        self.integrator = TSDFGlobalIntegrator(self.volume, keyframes, cfg, self.optimizer)

    def start(self):
        pass

    def shutdown(self):
        pass

    def on_after_backend_solve(self, factor_graph):
        """global_manager.py:213-226, followed by the pass the reference's two threads would make."""
        if not self.enabled:
            return
        self.integrator._integrate_new_keyframes()
        idx_tensor = getattr(factor_graph, "last_unique_kf_idx", None)
        if idx_tensor is None:
            return
        pin = int(config.get("local_opt", {}).get("pin", 1))
        indices = [int(i) for i in idx_tensor.cpu().tolist() if int(i) >= pin]
        if not indices:
            return
        self.integrator.mark_pose_update(indices)
        self.integrator._process_dirty_queue()
        self.optimizer.optimize_keyframes(indices, context="factor")
