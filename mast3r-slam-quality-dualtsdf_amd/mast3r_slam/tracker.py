"""Mirror of mast3r_slam/tracker.py (FrameTracker, lines 15-317): same class, method names and return
values.  Inference + matching + the whole <=50-iteration Sim3 Gauss-Newton run in libmslam_hip.so; the
quality-service submission (tracker.py:94-145) happens when a `quality_service` is attached (main.py:246), with
device tensors in the job instead of numpy copies."""
import numpy as np
import torch

import mslam_hip as _m
from lietorch_hip import Sim3
from mast3r_slam.config import config
from mast3r_slam.geometry import act_Sim3, constrain_points_to_ray, point_to_ray_dist
from mast3r_slam.mast3r_utils import mast3r_match_asymmetric


class FrameTracker:
    def __init__(self, model, frames, device):
        self.cfg = config["tracking"]
        self.model = model
        self.keyframes = frames
        self.device = device
        self.reset_idx_f2k()
        self._ws = None
        self._status = None
        self.quality_service = None
        self.last_kf_value = None

    def reset_idx_f2k(self):
        self.idx_f2k = None

    # ------------------------------------------------------------------
    def track(self, frame):
        """tracker.py:28-179 -> (new_kf, [Xk, Ck_avg, Xf, Cf_avg, Qkf, Qff], skipped)."""
        keyframe = self.keyframes.last_keyframe()
        idx_f2k, valid_match_k, Xff, Cff, Qff, Xkf, Ckf, Qkf = mast3r_match_asymmetric(
            self.model, frame, keyframe, idx_i2j_init=self.idx_f2k)
        self.idx_f2k = idx_f2k.clone()
        idx_f2k = idx_f2k[0]
        valid_match_k = valid_match_k[0]
        Qk = torch.sqrt(Qff[idx_f2k] * Qkf)
        frame.update_pointmap(Xff, Cff)

        use_calib = config["use_calib"]
        img_size = frame.img.shape[-2:]
        K = keyframe.K if use_calib else None
        Xf, Xk, T_WCf, T_WCk, Cf, Ck, meas_k, valid_meas_k = self.get_points_poses(
            frame, keyframe, idx_f2k, img_size, use_calib, K)

        valid_Cf = Cf > self.cfg["C_conf"]
        valid_Ck = Ck > self.cfg["C_conf"]
        valid_Q = Qk > self.cfg["Q_conf"]
        valid_opt = valid_match_k & valid_Cf & valid_Ck & valid_Q
        valid_kf = valid_match_k & valid_Q

        # ONE host read per tracked frame: the reference branches on the host four times (match fraction :72-75, solver
        # failure :91-93, the two fractions of the keyframe decision :170-177).  Here the solver is enqueued
        # unconditionally (its result is dropped when the match-fraction gate fails) and the five scalars come back in
        # one small copy; the unique count is a scatter instead of torch.unique (no second synchronisation).
        match_frac = valid_opt.float().mean()
        if not use_calib:
            T_WCf, T_CkCf, status = self._run_async(False, Xf, Xk, T_WCf, T_WCk, Qk, valid_opt, None, None, chunked=True)
        else:
            T_WCf, T_CkCf, status = self._run_async(True, Xf, Xk, T_WCf, T_WCk, Qk, valid_opt, K, img_size, chunked=True)
        hits = torch.zeros(valid_kf.numel(), dtype=torch.int32, device=idx_f2k.device)
        hits.index_add_(0, idx_f2k, valid_match_k[:, 0].to(torch.int32))
        pack = lambda st: torch.stack((match_frac, st[1].float(), st[2].float(), valid_kf.float().mean(),
                                       (hits > 0).float().mean(), st[0].float())).cpu()
        verdict = pack(status)
        if float(verdict[0]) >= self.cfg["min_match_frac"] and int(verdict[5]) == 0 and int(verdict[2]) == 0 \
                and int(verdict[1]) < int(self.cfg["max_iters"]):
            T_WCf, T_CkCf, status = self._run_rest()     # rare: the loop needs more than the first chunk
            verdict = pack(status)
        self.last_iters = int(verdict[1])
        if float(verdict[0]) < self.cfg["min_match_frac"]:
            return False, [], True
        if int(verdict[2]) != 0:  # "Cholesky failed" (tracker.py:91-93)
            return False, [], True

        if self.quality_service is not None and not use_calib:   # tracker.py:94-145 (ray-distance residual form)
            Xf_g = Xf[idx_f2k]                                    # the reference's Xf is the gathered one (:181-206)
            rd_k = point_to_ray_dist(Xk, jacobian=False)
            rd_f = point_to_ray_dist(act_Sim3(T_CkCf, Xf_g, jacobian=False), jacobian=False)
            vec = T_CkCf.data.view(-1, 8)
            w = vec[..., 6].clamp(-1.0, 1.0).abs()
            self.quality_service.submit({
                "kf_id": int(len(self.keyframes) - 1), "frame_id": int(keyframe.frame_id), "H": int(img_size[0]),
                "W": int(img_size[1]), "valid_kf": valid_kf.view(-1), "r_pix": torch.linalg.norm(rd_k - rd_f, dim=1),
                "Ck": Ck.view(-1), "Qk": Qk.view(-1), "t_norm": vec[..., :3].norm(dim=-1).mean(),
                "theta": (2.0 * torch.arccos(w)).mean()})

        frame.T_WC = T_WCf
        Xkk = T_CkCf.act(Xkf)
        keyframe.update_pointmap(Xkk, Ckf)
        self.keyframes[len(self.keyframes) - 1] = keyframe

        match_frac_k, unique_frac_f = float(verdict[3]), float(verdict[4])
        self.last_kf_value = min(match_frac_k, unique_frac_f)     # how far the keyframe rule is from firing
        new_kf = min(match_frac_k, unique_frac_f) < self.cfg["match_frac_thresh"]
        if new_kf:
            self.reset_idx_f2k()
        return (new_kf, [keyframe.X_canon, keyframe.get_average_conf(), frame.X_canon, frame.get_average_conf(),
                         Qkf, Qff], False)

    def get_points_poses(self, frame, keyframe, idx_f2k, img_size, use_calib, K=None):
        """tracker.py:181-206.  Unlike the reference this returns the UN-gathered frame points and
        the index (the gather happens inside the GN kernel); Cf is gathered as in the reference."""
        Xf, Xk = frame.X_canon, keyframe.X_canon
        Cf, Ck = frame.get_average_conf(), keyframe.get_average_conf()
        meas_k = valid_meas_k = None
        if use_calib:
            Xf = constrain_points_to_ray(img_size, Xf[None], K).squeeze(0)
            Xk = constrain_points_to_ray(img_size, Xk[None], K).squeeze(0)
        self._idx = idx_f2k
        return Xf, Xk, frame.T_WC, keyframe.T_WC, Cf[idx_f2k], Ck, meas_k, valid_meas_k

    # ------------------------------------------------------------------
    def _run(self, use_calib, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx=None):
        T_WCf_new, T_CkCf, status = self._run_async(use_calib, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx)
        st = status.cpu()   # synchronous form (opt_pose_* keep the reference's (T_WCf, T_CkCf, ok) return)
        self.last_iters = int(st[1])
        return T_WCf_new, T_CkCf, int(st[2]) == 0

    FIRST_CHUNK = 8   # iterations enqueued before the verdict is read; the rest only if the loop has not finished

    def _run_async(self, use_calib, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx=None, chunked=False):
        """Enqueues the GN loop; returns (T_WCf, T_CkCf, status) with `status` a device i32[8]
        ([done, iterations, failed, ...]) that the caller reads when it needs the verdict.  chunked=True enqueues only
        the first FIRST_CHUNK iterations (a tracked frame needs ~5; 50 launch pairs that exit at once would cost more
        host time than the solve): the caller checks `done` and calls _run_rest when the loop is still running."""
        cfg = self.cfg
        dev = Xf.device
        idx = self._idx if idx is None else idx
        n = Xk.shape[0]
        T_rel = (T_WCk.inv() * T_WCf).data.reshape(8).contiguous().clone()
        L = _m.lib()
        need = L.mslam_track_workspace_bytes(n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._status = torch.zeros(8, dtype=torch.int32, device=dev)
        h, w = (int(img_size[0]), int(img_size[1])) if img_size is not None else (0, 0)
        sa, sb = (cfg["sigma_pixel"], cfg["sigma_depth"]) if use_calib else (cfg["sigma_ray"], cfg["sigma_dist"])
        # the contiguous forms are named and kept in the job: _run_rest relaunches with the same pointers after other
        # allocations have happened on the stream (a temporary .contiguous() copy would be freed at once)
        Xf_c, Xk_c, idx_c = Xf.contiguous(), Xk.contiguous(), idx.contiguous()
        Qk_c, valid_c = Qk.reshape(-1).contiguous(), valid.reshape(-1).contiguous()
        K_c = K.contiguous() if use_calib else None
        self._job = dict(args=(int(use_calib), _m.ptr(T_rel), _m.ptr(Xf_c), _m.ptr(Xk_c), _m.ptr(idx_c), _m.ptr(Qk_c),
                               _m.ptr(valid_c), n, _m.ptr(K_c) if use_calib else 0,
                               w, h, float(sa), float(sb), float(cfg["huber"]), int(cfg["pixel_border"]),
                               float(cfg["depth_eps"])),
                         keep=(T_rel, Xf_c, Xk_c, idx_c, Qk_c, valid_c, K_c), T_rel=T_rel, T_WCk=T_WCk)
        last = min(self.FIRST_CHUNK, int(cfg["max_iters"])) if chunked else int(cfg["max_iters"])
        return self._enqueue(0, last)

    def _enqueue(self, first, last):
        cfg, job = self.cfg, self._job
        rc = _m.lib().mslam_track_pose(*job["args"], int(first), int(last), float(cfg["rel_error"]),
                                       float(cfg["delta_norm"]), _m.ptr(self._status), _m.ptr(self._ws),
                                       self._ws.numel(), _m.stream_ptr())
        _m.check(rc, "track_pose")
        T_CkCf = Sim3(job["T_rel"].reshape(1, 8))
        return job["T_WCk"] * T_CkCf, T_CkCf, self._status

    def _run_rest(self):
        """The iterations behind the first chunk (same loop state, same results as one uninterrupted loop)."""
        return self._enqueue(min(self.FIRST_CHUNK, int(self.cfg["max_iters"])), int(self.cfg["max_iters"]))

    def opt_pose_ray_dist_sim3(self, Xf, Xk, T_WCf, T_WCk, Qk, valid, idx=None):
        """tracker.py:225-266 -> (T_WCf, T_CkCf, ok)."""
        return self._run(False, Xf, Xk, T_WCf, T_WCk, Qk, valid, None, None, idx)

    def opt_pose_calib_sim3(self, Xf, Xk, T_WCf, T_WCk, Qk, valid, meas_k, valid_meas_k, K, img_size, idx=None):
        """tracker.py:268-318 -> (T_WCf, T_CkCf, ok).  meas_k / valid_meas_k are derived from Xk inside
        the kernel (pixel grid + log depth, tracker.py:197-203)."""
        return self._run(True, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx)
