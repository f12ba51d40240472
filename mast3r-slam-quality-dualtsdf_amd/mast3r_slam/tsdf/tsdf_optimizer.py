"""Mirror of mast3r_slam/tsdf/tsdf_optimizer.py (TSDFPoseOptimizer, lines 9-124): Sim3 pose
refinement of a keyframe against the global TSDF.  The per-point python loops (query, Jacobian,
outer products) and the 7x7 solve run as two kernels per iteration with no host round trip."""
import torch

import mslam_hip as _m
from lietorch_hip import Sim3


class TSDFPoseOptimizer:
    def __init__(self, volume, keyframes, cfg, use_calib, device):
        self.volume = volume
        self.keyframes = keyframes
        self.cfg = cfg
        self.use_calib = use_calib
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.samples_per_kf = int(cfg.get("samples_per_kf", 2000))
        self.min_conf = float(cfg.get("min_confidence", 0.05))
        self.max_iterations = int(cfg.get("max_iterations", 3))
        self.lambda_tsdf = float(cfg.get("lambda", 0.1))
        self.damping = float(cfg.get("damping", 1.0e-4))
        self.pre_icp_iters = int(cfg.get("pre_icp_iters", 0))
        self._ws = torch.empty(64 * 36 * 8, dtype=torch.uint8, device=volume.device)

    # ------------------------------------------------------------------ reference method surface
    def pre_refine(self, kf_idx):
        """tsdf_optimizer.py:29-37."""
        if self.pre_icp_iters <= 0:
            return
        self._optimize_single(kf_idx, iterations=self.pre_icp_iters, sample_override=min(self.samples_per_kf // 2, 1000))

    def optimize_keyframes(self, indices, context="factor"):
        """tsdf_optimizer.py:39-43."""
        for idx in indices or ():
            self._optimize_single(idx, iterations=self.max_iterations)

    def _optimize_single(self, idx, iterations, sample_override=0, log_prefix="[TSDF]"):
        """tsdf_optimizer.py:46-92: sample <= samples_per_kf points with C > min_confidence (torch.randperm,
        as the reference), iterate pose <- exp(delta) * pose against the volume, write T_WC back."""
        if idx >= len(self.keyframes) or iterations <= 0:
            return
        frame = self.keyframes[idx]
        points = frame.X_canon.detach().reshape(-1, 3)
        conf = frame.C.detach().reshape(-1)
        valid_idx = torch.nonzero(conf > self.min_conf).view(-1)
        if valid_idx.numel() == 0:
            return
        max_samples = sample_override if sample_override > 0 else self.samples_per_kf
        count = min(max_samples, valid_idx.numel())
        choice = valid_idx[torch.randperm(valid_idx.numel(), device=valid_idx.device)[:count]]
        pose = self.refine_pose(Sim3(frame.T_WC.data.clone()), points[choice], conf[choice], iterations=iterations)
        # write back THROUGH the store (tsdf_optimizer.py:88-92 assigns keyframes.T_WC[idx] under the lock): with the
        # SharedKeyframes buffers `frame` is a temporary of views and rebinding its attribute would be lost
        self.keyframes.update_T_WCs(pose, torch.tensor([idx]))

    def normal_equations(self, points_world, conf):
        """_build_linear_system + _accumulate_system (tsdf_optimizer.py:94-116) for world points:
        returns (H f64[7,7], b f64[7], used i32) device tensors.  Voxel-sharded volume: collective (owner-computes
        look-up, one all-reduce), same bits as one table."""
        v = self.volume
        pts = v._dev(points_world, torch.float32).reshape(-1, 3)
        cf = v._dev(conf, torch.float32).reshape(-1)
        H = torch.zeros((7, 7), dtype=torch.float64, device=v.device)
        b = torch.zeros(7, dtype=torch.float64, device=v.device)
        used = torch.zeros(1, dtype=torch.int32, device=v.device)
        L = _m.lib()
        if v._collective:
            if v._driver:
                from mast3r_slam import shard as sh

                with v.channel.lock:
                    v.channel.announce(sh.OP_TSDF_NEQ, [pts.shape[0]])
                    v.channel.bcast(torch.cat((pts, cf[:, None]), dim=1).contiguous())
            lk = v.lookup7(pts)
            rc = L.mslam_tsdf_pose_step_lookup(
                _m.ptr(lk), _m.ptr(pts), _m.ptr(cf), pts.shape[0], 0, 0, v.voxel_size, v.min_weight, self.lambda_tsdf,
                self.damping, 0, _m.ptr(H), _m.ptr(b), _m.ptr(used), _m.ptr(self._ws), self._ws.numel(), _m.stream_ptr())
            _m.check(rc, "tsdf_pose_step_lookup")
            return H, b, used
        rc = L.mslam_tsdf_pose_step(
            _m.ptr(v._table), v.capacity, _m.ptr(pts), _m.ptr(cf), pts.shape[0], 0, 0, v.voxel_size, v.min_weight,
            self.lambda_tsdf, self.damping, 0, _m.ptr(H), _m.ptr(b), _m.ptr(used), _m.ptr(self._ws),
            self._ws.numel(), _m.stream_ptr())
        _m.check(rc, "tsdf_pose_step")
        return H, b, used

    def refine_pose(self, pose: Sim3, pts_cam, conf, iterations=None):
        """The iteration loop of _optimize_single (tsdf_optimizer.py:76-86) for already-sampled
        camera-frame points: pose <- exp(delta) * pose, `iterations` times.  Returns the new Sim3.  Voxel-sharded
        volume: collective; per iteration every rank looks the moved points up in its shard, one all-reduce (336 KB at
        2 000 samples), and every rank takes the same step (identical bits, no pose broadcast)."""
        v = self.volume
        iterations = self.max_iterations if iterations is None else iterations
        pts = v._dev(pts_cam, torch.float32).reshape(-1, 3)
        cf = v._dev(conf, torch.float32).reshape(-1)
        data = pose.data.reshape(8).to(device=v.device, dtype=torch.float32).clone()
        L = _m.lib()
        if v._collective:
            n = pts.shape[0]
            if v._driver:
                from mast3r_slam import shard as sh

                pk = torch.empty((n + 2, 4), dtype=torch.float32, device=v.device)
                pk[:n, :3], pk[:n, 3], pk[n], pk[n + 1] = pts, cf, data[:4], data[4:]
                with v.channel.lock:
                    v.channel.announce(sh.OP_TSDF_REFINE, [n, int(iterations)])
                    v.channel.bcast(pk)
            for _ in range(iterations):
                lk = v.lookup7(pts, pose=data)
                rc = L.mslam_tsdf_pose_step_lookup(
                    _m.ptr(lk), _m.ptr(pts), _m.ptr(cf), n, _m.ptr(data), 1, v.voxel_size, v.min_weight, self.lambda_tsdf,
                    self.damping, 1, 0, 0, 0, _m.ptr(self._ws), self._ws.numel(), _m.stream_ptr())
                _m.check(rc, "tsdf_pose_step_lookup")
            return Sim3(data.reshape(1, 8))
        for _ in range(iterations):
            rc = L.mslam_tsdf_pose_step(
                _m.ptr(v._table), v.capacity, _m.ptr(pts), _m.ptr(cf), pts.shape[0], _m.ptr(data), 1, v.voxel_size,
                v.min_weight, self.lambda_tsdf, self.damping, 1, 0, 0, 0, _m.ptr(self._ws), self._ws.numel(),
                _m.stream_ptr())
            _m.check(rc, "tsdf_pose_step")
        return Sim3(data.reshape(1, 8))
