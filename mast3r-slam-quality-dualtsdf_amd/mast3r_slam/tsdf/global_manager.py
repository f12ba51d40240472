"""Mirror of mast3r_slam/tsdf/global_manager.py: TSDFGlobalIntegrator (lines 16-114) and TSDFGlobalManager
(lines 177-226), same constructor signatures and method names.  The reference runs the integrator and the pose
optimiser as daemon threads that poll every 0.1 s; here one synchronous pass of each runs inside
`on_after_backend_solve` on the caller's stream (bench.py / SlamSystem call it from the backend), so the order
integrate-new -> re-integrate-updated -> optimise is fixed instead of depending on thread timing.

Budgets.  In the reference the work a backend solve queues (re-fuse EVERY optimised keyframe, refine EVERY optimised
pose) is drained by the two threads at whatever rate they manage: the optimiser thread takes `max_opt_batch` (1)
keyframe per 0.15 s cycle (global_manager.py:131-160), the integrator what its thread time allows, both queues are
bounded (256) with de-duplication and whatever is still queued at shutdown is dropped (join timeout 2 s).  The
synchronous form keeps the queues (FIFO, de-duplicated, bounded) and spends a bounded budget per backend solve:
`sync_reintegrate_per_solve` re-fusions and `sync_optimize_per_solve` pose refinements (default 4 each; 0 = drain
completely), so the cost of a solve does not grow with the number of keyframes while every keyframe is still
revisited in turn."""
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

    def _process_dirty_queue(self, budget=0):
        """global_manager.py:71-79: keyframes whose pose the backend moved are fused again at the new pose
        (`budget` > 0: at most that many per call, the rest stay queued)."""
        done = 0
        while self.pending and (budget <= 0 or done < budget):
            idx = self.pending.pop(0)
            if idx < len(self.keyframes):
                self._integrate_single(idx)
                done += 1

    def run_once(self):
        """One pass of the reference thread's loop body (global_manager.py:57-60)."""
        self._integrate_new_keyframes()
        self._process_dirty_queue()

    def snapshot(self, idx):
        """The copies _integrate_single takes under the keyframes lock (global_manager.py:82-88): X, C, T_WC."""
        if idx >= len(self.keyframes):
            return None
        frame = self.keyframes[idx]
        if frame.X_canon.is_cuda:      # allocated on the tracking stream, which may replace the keyframe while the clones
            cur = torch.cuda.current_stream(frame.X_canon.device)   # below are still queued on this one
            for t in (frame.X_canon, frame.C, frame.T_WC.data):
                t.record_stream(cur)
        return (frame.X_canon.detach().clone(), frame.C.detach().clone(), frame.T_WC.data.clone())

    def _integrate_single(self, idx):
        """global_manager.py:81-106: random subset (<= max_points_per_kf) of the points with C > min_confidence,
        moved to the world frame with T_WC, fused with the camera centre as ray origin.  Everything stays on
        the device (the reference round-trips through numpy because its volume is a python dict)."""
        self._integrate_snapshot(self.snapshot(idx))

    def _integrate_snapshot(self, snap):
        if snap is None:
            return
        X_canon, C, T_data = snap
        points = X_canon.reshape(-1, 3)
        conf = C.reshape(-1)
        valid_idx = torch.nonzero(conf > self.min_conf).view(-1)
        if valid_idx.numel() == 0:
            return
        count = min(valid_idx.numel(), self.max_points)
        choice = valid_idx[torch.randperm(valid_idx.numel(), device=valid_idx.device)[:count]]
        pose = Sim3(T_data)
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

    def __init__(self, keyframes, cfg, use_calib, device, channel=None, shard_id=0, num_shards=1, group=None):
        """`channel` (mast3r_slam/shard.py, driver rank of a sharded session): this rank keeps shard 0 of the voxels, the
        shard ranks the others, every volume call is announced.  `shard_id` / `num_shards` / `group` without a channel:
        every rank runs its own manager on the same keyframes (SPMD)."""
        from .global_volume import TSDFVolume

        from .tsdf_optimizer import TSDFPoseOptimizer

        self.enabled = bool(cfg.get("enabled", False))
        self.keyframes = keyframes
        self.cfg = cfg
        self.volume = TSDFVolume(voxel_size=cfg.get("voxel_size", 0.03), truncation=cfg.get("trunc_dist", 0.12),
                                 max_weight=cfg.get("max_weight", 100.0), min_weight=cfg.get("min_tsdf_weight", 1.0e-3),
                                 capacity=int(cfg.get("hash_capacity", 1 << 22)), device=device,
                                 shard_id=channel.rank if channel is not None else shard_id,
                                 num_shards=channel.world if channel is not None else num_shards, group=group,
                                 channel=channel)
        self.optimizer = TSDFPoseOptimizer(self.volume, keyframes, cfg, use_calib, device)
        self.integrator = TSDFGlobalIntegrator(self.volume, keyframes, cfg, self.optimizer)
        self.reintegrate_budget = int(cfg.get("sync_reintegrate_per_solve", 4))
        self.optimize_budget = int(cfg.get("sync_optimize_per_solve", 4))
        self.max_opt_pending = int(cfg.get("opt_queue", cfg.get("reintegration_queue", 256)))
        self.opt_pending = []     # TSDFGlobalOptThread.queue + pending (global_manager.py:121-143)

    def start(self):
        pass

    def _maintain(self, n_fusions):
        """One small D2H read per solve, BEFORE its work: samples dropped by the previous solve raise; the table grows so
        that everything this solve may insert (`n_fusions` keyframes, every in-band sample a new voxel) still leaves
        it half empty - the table cannot grow in the middle of an integrate."""
        v = self.volume
        band = int(2.0 * v.truncation / (0.5 * v.voxel_size)) + 4
        self.volume.maintain(reserve=n_fusions * self.integrator.max_points * band)

    def shutdown(self):
        if self.enabled:
            self.volume.maintain()   # samples dropped by the last solve are reported here

    def on_after_backend_solve(self, factor_graph):
        """global_manager.py:213-226, followed by the pass the reference's two threads would make."""
        self.execute(self.plan(factor_graph))

    # The hook in two halves for a threaded owner: plan() does the queue bookkeeping and takes the copies of the keyframe
    # tensors the fusions need (what the reference does under keyframes.lock) - no host synchronisation; execute() does
    # the device work on those copies.  The pose optimiser still reads and writes the store directly, so a threaded
    # owner must run execute() where that is safe whenever `sync_optimize_per_solve` leaves it work (max_iterations > 0).
    def plan(self, factor_graph):
        if not self.enabled:
            return None
        integ = self.integrator
        new = list(range(integ.next_idx, len(self.keyframes)))
        todo = [("new", i, integ.snapshot(i)) for i in new]
        integ.next_idx = len(self.keyframes)
        batch = []
        idx_tensor = getattr(factor_graph, "last_unique_kf_idx", None)
        if idx_tensor is not None:
            pin = int(config.get("local_opt", {}).get("pin", 1))
            indices = [int(i) for i in idx_tensor.tolist() if int(i) >= pin]
            if indices:
                integ.mark_pose_update(indices)
                done = 0
                while integ.pending and (self.reintegrate_budget <= 0 or done < self.reintegrate_budget):
                    i = integ.pending.pop(0)
                    if i < len(self.keyframes):
                        todo.append(("dirty", i, integ.snapshot(i)))
                        done += 1
                for i in indices:      # TSDFGlobalOptThread.enqueue
                    if i not in self.opt_pending and len(self.opt_pending) < self.max_opt_pending:
                        self.opt_pending.append(i)
                n = len(self.opt_pending) if self.optimize_budget <= 0 else min(self.optimize_budget, len(self.opt_pending))
                batch, self.opt_pending = self.opt_pending[:n], self.opt_pending[n:]
        return dict(todo=todo, optimize=batch)

    def retarget(self, plan, kf_idx_host, pose_data):
        """Threaded owner: plan() ran BEFORE the solve (it copies keyframe data under the hand-over lock), the fusions
        must happen at the poses the solve produced (the inline order, and the reference: the integrator thread reads
        T_WC after the backend wrote it, global_manager.py:82-88).  Replaces the pose of every planned fusion whose
        keyframe the solve held; `kf_idx_host` = the solve's keyframe ids (host), `pose_data` = its (P, 8) poses."""
        if plan is None:
            return
        row = {int(k): r for r, k in enumerate(kf_idx_host.tolist())}
        todo = []
        for kind, i, snap in plan["todo"]:
            if snap is not None and i in row:
                snap = (snap[0], snap[1], pose_data[row[i]].reshape(1, 8).clone())
            todo.append((kind, i, snap))
        plan["todo"] = todo

    def has_pose_refinement(self, plan):
        """True when execute_optimize(plan) would touch keyframe poses in the store."""
        if plan is None:
            return False
        opt = self.optimizer
        pre = opt.pre_icp_iters > 0 and any(kind == "new" for kind, _, _ in plan["todo"])
        return pre or (opt.max_iterations > 0 and len(plan["optimize"]) > 0)

    def execute(self, plan):
        if plan is None:
            return
        self._maintain(len(plan["todo"]))
        for kind, i, snap in plan["todo"]:
            self.integrator._integrate_snapshot(snap)
            if kind == "new":
                self.optimizer.pre_refine(i)
        self.optimizer.optimize_keyframes(plan["optimize"], context="factor")
