"""Mirror of mast3r_slam/tracker.py (FrameTracker, lines 15-317): same class, method names and return
values.  Inference + matching + the whole <=50-iteration Sim3 Gauss-Newton run in libmslam_hip.so; the
quality-service submission (tracker.py:94-145) happens when a `quality_service` is attached (main.py:246), with
device tensors in the job instead of numpy copies."""
import copy
import time

import numpy as np
import torch

import mslam_hip as _m
from lietorch_hip import Sim3
from mast3r_slam.config import config
from mast3r_slam.geometry import act_Sim3, constrain_points_to_ray, point_to_ray_dist
from mast3r_slam.mast3r_utils import mast3r_match_asymmetric


class _Pending:
    """One tracked frame between track_begin (everything enqueued, effects applied optimistically) and track_finish."""
    __slots__ = ("frame", "init_T", "prev_idx", "prev_shadow", "stash", "base", "shadow", "slot", "job", "verdict",
                 "event", "quality", "T_WCf", "T_CkCf", "T_WCk", "Xkf", "Ckf", "Qkf", "Qff", "status", "n_points", "replayed",
                 "kind", "vals", "new_kf")


class FrameTracker:
    IN_PLACE_MODES = ("indep_conf",)      # Frame.update_pointmap edits X_canon / C in place in these modes

    def __init__(self, model, frames, device):
        self.cfg = config["tracking"]
        self.model = model
        self.keyframes = frames
        self.device = device
        self.reset_idx_f2k()
        self.N_SLOTS = 4           # solver loop states in flight: the frame being resolved + up to 3 begun behind it
        self._slots = [dict(ws=None, status=None, host=None, prep=None) for _ in range(self.N_SLOTS)]
        self._slot = 0
        self._shadow = None        # fused state of the current keyframe that has not been written to the store yet
        self.quality_service = None
        self.last_kf_value = None
        self.verdict_wait_s = 0.0  # host time spent waiting for verdicts (the one blocking read per tracked frame)

    def reset_idx_f2k(self):
        self.idx_f2k = None

    # ------------------------------------------------------------------
    def track(self, frame):
        """tracker.py:28-179 -> (new_kf, [Xk, Ck_avg, Xf, Cf_avg, Qkf, Qff], skipped)."""
        h = self.track_begin(frame)
        self.track_resolve(h)
        return self.track_finish(h)

    # The three phases of track().  A caller that pipelines frames (SlamSystem.run) enqueues frame f+1 (track_begin) BEFORE
    # it reads frame f's verdict (track_resolve): the one host read per frame then no longer idles the device.
    # track_begin applies every effect of a tracked frame optimistically ("tracked, keyframe stays", true for ~88 % of
    # the frames) but keeps the fused keyframe in a private shadow copy; track_finish writes it to the store, or
    # drops it when the frame was skipped; rollback() undoes a begin whose premise (the previous frame's outcome) failed.
    def track_begin(self, frame):
        h = _Pending()
        h.frame, h.replayed = frame, False
        h.init_T, h.prev_idx, h.prev_shadow = frame.T_WC, self.idx_f2k, self._shadow
        h.stash = getattr(frame, "decoded", None)
        stored = self.keyframes.last_keyframe()
        keyframe = self._shadow if self._shadow is not None else stored
        idx_f2k, valid_match_k, Xff, Cff, Qff, Xkf, Ckf, Qkf = mast3r_match_asymmetric(
            self.model, frame, keyframe, idx_i2j_init=self.idx_f2k)
        self.idx_f2k = idx_f2k.clone()
        idx_f2k = idx_f2k[0]
        valid_match_k = valid_match_k[0]
        frame.update_pointmap(Xff, Cff)

        use_calib = config["use_calib"]
        img_size = frame.img.shape[-2:]
        K = keyframe.K if use_calib else None
        T_WCk = stored.T_WC            # poses live in the store (the backend's write-backs land there)
        Xf, Xk = frame.X_canon, keyframe.X_canon       # get_points_poses (tracker.py:181-206) without its tensor ops:
        if use_calib:                                  # the gathers, C / N and the masks below happen in ONE launch
            Xf = constrain_points_to_ray(img_size, Xf[None], K).squeeze(0)
            Xk = constrain_points_to_ray(img_size, Xk[None], K).squeeze(0)
        self._idx = idx_f2k

        # ONE host read per tracked frame: the reference branches on the host four times (match fraction :72-75, solver
        # failure :91-93, the two fractions of the keyframe decision :170-177).  Here the solver is enqueued
        # unconditionally (its result is dropped when the match-fraction gate fails) and the six scalars come back in
        # one small copy into pinned memory behind an event.  Qk, the masks (tracker.py:61-70), their fractions, the
        # unique-match count and T_CkCf = T_WCk^-1 * T_WCf are one launch (mslam_track_prepare) instead of ~30 tensor
        # ops with a launch bubble behind each.
        h.slot = self._slot
        self._slot = (self._slot + 1) % self.N_SLOTS
        self._cur = h
        Qk, Ck, valid_opt, valid_kf, T_rel = self._prepare(h, frame, keyframe, idx_f2k, valid_match_k, Qff, Qkf,
                                                           T_WCk, frame.T_WC)
        T_WCf, T_CkCf, status = self._run_async(use_calib, Xf, Xk, None, T_WCk, Qk, valid_opt, K if use_calib else None,
                                                img_size if use_calib else None, chunked=True, T_rel=T_rel, pose=False)
        h.job, h.T_WCk = getattr(self, "_job", None), T_WCk
        h.status = status
        self._post_verdict(h)
        h.quality = None
        if self.quality_service is not None and not use_calib:   # tracker.py:94-145 (ray-distance residual form)
            # the residual map is needed only if this job is still the keyframe's newest one when the service computes
            # (a newer job replaces a queued one): the inputs ride along, the tensor expressions run in the service
            h.quality = dict(Xf=Xf, idx=idx_f2k, Xk=Xk, valid_kf=valid_kf.view(-1), Ck=Ck.view(-1), Qk=Qk.view(-1),
                             kf_id=int(len(self.keyframes) - 1), frame_id=int(keyframe.frame_id), H=int(img_size[0]),
                             W=int(img_size[1]))
        h.base, h.Xkf, h.Ckf, h.Qkf, h.Qff = keyframe, Xkf, Ckf, Qkf, Qff
        self._apply(h, T_WCf, T_CkCf)
        return h

    def _prepare(self, h, frame, keyframe, idx_f2k, valid_match_k, Qff, Qkf, T_WCk, T_WCf):
        """tracker.py:61-70 + the fractions of :72-75 / :170-177 + T_CkCf -> (Qk, Ck, valid_opt, valid_kf, T_rel);
        the three counts stay on the device (slot workspace) for _post_verdict."""
        n = int(keyframe.X_canon.shape[0])
        dev = idx_f2k.device
        L = _m.lib()
        sl = self._slots[h.slot]
        need = L.mslam_track_prepare_workspace_bytes(n)
        if sl.get("prep") is None or sl["prep"].numel() < need:
            sl["prep"] = torch.empty(need, dtype=torch.uint8, device=dev)
        Qk = torch.empty((n, 1), dtype=torch.float32, device=dev)
        Ck = torch.empty((n, 1), dtype=torch.float32, device=dev)
        valid_opt = torch.empty((n, 1), dtype=torch.bool, device=dev)
        valid_kf = torch.empty((n, 1), dtype=torch.bool, device=dev)
        T_rel = torch.empty(8, dtype=torch.float32, device=dev)
        idx_c, vm_c = idx_f2k.contiguous(), valid_match_k.contiguous()
        Qff_c, Qkf_c = Qff.contiguous(), Qkf.contiguous()
        Cf_c, Ck_c = frame.C.contiguous(), keyframe.C.contiguous()
        Tk_c, Tf_c = T_WCk.data.reshape(8).contiguous(), T_WCf.data.reshape(8).contiguous()
        _m.require_dtype(idx_c, torch.int64, "idx_f2k")
        _m.require_dtype(vm_c, torch.bool, "valid_match_k")
        # Frame.get_average_conf is C / N with N a python int: on the device that is C * (1 / N) in fp32
        inv_nf = float(np.float32(1.0) / np.float32(frame.N))
        inv_nk = float(np.float32(1.0) / np.float32(keyframe.N))
        rc = L.mslam_track_prepare(_m.ptr(idx_c), _m.ptr(vm_c), _m.ptr(Qff_c), _m.ptr(Qkf_c), _m.ptr(Cf_c), inv_nf,
                                   _m.ptr(Ck_c), inv_nk, float(self.cfg["C_conf"]), float(self.cfg["Q_conf"]), n,
                                   _m.ptr(Tk_c), _m.ptr(Tf_c), _m.ptr(Qk), _m.ptr(Ck), _m.ptr(valid_opt),
                                   _m.ptr(valid_kf), _m.ptr(T_rel), _m.ptr(sl["prep"]), sl["prep"].numel(),
                                   _m.stream_ptr())
        _m.check(rc, "track_prepare")
        h.n_points = n
        return Qk, Ck, valid_opt, valid_kf, T_rel

    def _post_verdict(self, h):
        """The six scalars of the verdict -> pinned host memory, asynchronously, with an event behind the copy."""
        sl = self._slots[h.slot]
        dev = torch.empty(6, dtype=torch.float32, device=h.status.device)
        rc = _m.lib().mslam_track_verdict(_m.ptr(sl["prep"]), _m.ptr(h.status), int(h.n_points), _m.ptr(dev),
                                          _m.stream_ptr())
        _m.check(rc, "track_verdict")
        if sl["host"] is None:
            sl["host"] = torch.empty(6, dtype=torch.float32).pin_memory()
        sl["host"].copy_(dev, non_blocking=True)
        h.event = torch.cuda.Event()
        h.event.record()
        h.verdict = sl["host"]

    def _apply(self, h, T_WCf, T_CkCf):
        """Optimistic effects of a tracked frame (tracker.py:147-168): the frame's pose, the keyframe's pointmap fused
        with the frame's view of it - into a shadow copy, the store is written by track_finish.  In the default
        filtering mode ('weighted_pointmap', frame.py:72-75) pose product, act and fusion are one launch."""
        h.T_CkCf = T_CkCf
        base = h.base
        shadow = copy.copy(base)
        shadow._in_store, shadow._replaced_by = False, None      # (the copy inherits the marks of a committed base)
        T_WCk = h.T_WCk
        if config["tracking"]["filtering_mode"] == "weighted_pointmap" and base.N > 0:
            n = int(base.X_canon.shape[0])
            dev = base.X_canon.device
            T_out = torch.empty((1, 8), dtype=torch.float32, device=dev)
            X_new, C_new = torch.empty_like(base.X_canon), torch.empty_like(base.C)
            Tk_c, Tr_c = T_WCk.data.reshape(8).contiguous(), T_CkCf.data.reshape(8).contiguous()
            Xkf_c, Ckf_c = h.Xkf.contiguous(), h.Ckf.contiguous()
            Xc_c, Cc_c = base.X_canon.contiguous(), base.C.contiguous()
            assert X_new.is_contiguous() and C_new.is_contiguous()
            rc = _m.lib().mslam_track_fuse(_m.ptr(Tk_c), _m.ptr(Tr_c), _m.ptr(Xkf_c), _m.ptr(Ckf_c), _m.ptr(Xc_c),
                                           _m.ptr(Cc_c), n, _m.ptr(T_out), _m.ptr(X_new), _m.ptr(C_new), _m.stream_ptr())
            _m.check(rc, "track_fuse")
            if T_WCf is None:                  # (a caller that solved the pose itself passes it)
                T_WCf = Sim3(T_out)
            shadow.X_canon, shadow.C = X_new, C_new
            shadow.N, shadow.N_updates = base.N + 1, base.N_updates + 1
        else:
            if T_WCf is None:
                T_WCf = T_WCk * T_CkCf
            Xkk = T_CkCf.act(h.Xkf)
            if config["tracking"]["filtering_mode"] in self.IN_PLACE_MODES:
                shadow.X_canon, shadow.C = shadow.X_canon.clone(), shadow.C.clone()
            shadow.update_pointmap(Xkk, h.Ckf)
        h.T_WCf = T_WCf
        h.frame.T_WC = T_WCf
        h.shadow = shadow
        self._shadow = shadow

    def track_resolve(self, h):
        """Reads the verdict (the one host wait of a tracked frame) -> "ok" or "skip".  When the solver needed more than
        the first chunk of iterations (rare) the rest runs now and the optimistic effects are redone from its result;
        `h.replayed` then tells a pipelining caller that a frame begun on top of this one saw stale inputs."""
        t_wait = time.perf_counter()
        h.event.synchronize()
        self.verdict_wait_s += time.perf_counter() - t_wait
        v = h.verdict.tolist()
        if v[0] >= self.cfg["min_match_frac"] and int(v[5]) == 0 and int(v[2]) == 0 and int(v[1]) < int(self.cfg["max_iters"]):
            self._cur = h
            _, T_CkCf, h.status = self._run_rest()         # rare: the loop needs more than the first chunk
            self._post_verdict(h)
            h.event.synchronize()
            v = h.verdict.tolist()
            stale = h.shadow
            later = self._shadow if self._shadow is not stale else None
            self._apply(h, None, T_CkCf)
            stale._replaced_by = h.shadow  # an undo that restores the stale shadow gets the redone one (_live)
            if later is not None:          # a frame was begun on top of the stale shadow: its caller rolls it back
                self._shadow = later
            h.replayed = True
        h.vals = v
        self.last_iters = int(v[1])
        skip = v[0] < self.cfg["min_match_frac"] or int(v[2]) != 0     # tracker.py:72-75 / "Cholesky failed" :91-93
        h.kind = "skip" if skip else "ok"
        h.new_kf = (not skip) and min(float(v[3]), float(v[4])) < self.cfg["match_frac_thresh"]     # tracker.py:170-177
        return h.kind

    def track_finish(self, h):
        """Host-side consequences of the verdict -> track()'s return value."""
        frame = h.frame
        if h.kind == "skip":
            frame.T_WC = h.init_T
            if self._shadow is h.shadow:
                self._shadow = self._live(h.prev_shadow)
            return False, [], True
        if h.quality is not None:
            q = h.quality
            T_CkCf = h.T_CkCf

            def residual_map(q=q, T=T_CkCf):       # tracker.py:94-127, evaluated by the service when it computes the job
                Xf_g = q["Xf"][q["idx"]]                              # the reference's Xf is the gathered one (:181-206)
                rd_k = point_to_ray_dist(q["Xk"], jacobian=False)
                rd_f = point_to_ray_dist(act_Sim3(T, Xf_g, jacobian=False), jacobian=False)
                vec = T.data.view(-1, 8)
                w = vec[..., 6].clamp(-1.0, 1.0).abs()
                return {"r_pix": torch.linalg.norm(rd_k - rd_f, dim=1), "t_norm": vec[..., :3].norm(dim=-1).mean(),
                        "theta": (2.0 * torch.arccos(w)).mean()}

            self.quality_service.submit({
                "kf_id": q["kf_id"], "frame_id": q["frame_id"], "H": q["H"], "W": q["W"], "valid_kf": q["valid_kf"],
                "Ck": q["Ck"], "Qk": q["Qk"], "lazy": residual_map, "keep": (q["Xf"], q["idx"], q["Xk"], T_CkCf.data)})
        keyframe = h.shadow
        keyframe.T_WC = self.keyframes.last_keyframe().T_WC      # never write a pose back that a solve has replaced since
        self.keyframes[len(self.keyframes) - 1] = keyframe
        keyframe._in_store = True                                 # a later frame's undo must not resurrect it as a shadow
        if self._shadow is h.shadow:
            self._shadow = None                                   # the store holds it now
        match_frac_k, unique_frac_f = float(h.vals[3]), float(h.vals[4])
        self.last_kf_value = min(match_frac_k, unique_frac_f)     # how far the keyframe rule is from firing
        new_kf = h.new_kf
        if new_kf:
            self.reset_idx_f2k()
            self._shadow = None
        return (new_kf, [keyframe.X_canon, keyframe.get_average_conf(), frame.X_canon, frame.get_average_conf(),
                         h.Qkf, h.Qff], False)

    @staticmethod
    def _live(shadow):
        """A shadow that has been written to the store since is no shadow any more (the store's keyframe is read)."""
        while shadow is not None and getattr(shadow, "_replaced_by", None) is not None:
            shadow = shadow._replaced_by
        return None if shadow is None or getattr(shadow, "_in_store", False) else shadow

    def rollback(self, h):
        """Undo track_begin(h) (nothing of it has reached the store): the tracker's chain state, the frame's own fields
        and the group-decode result it consumed, so that the frame can be begun again."""
        self.idx_f2k, self._shadow = h.prev_idx, self._live(h.prev_shadow)
        f = h.frame
        f.T_WC, f.X_canon, f.C, f.N, f.N_updates = h.init_T, None, None, 0, 0
        if h.stash is not None:
            f.decoded = h.stash

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

    def _run_async(self, use_calib, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx=None, chunked=False, T_rel=None,
                   pose=True):
        """Enqueues the GN loop; returns (T_WCf, T_CkCf, status) with `status` a device i32[8]
        ([done, iterations, failed, ...]) that the caller reads when it needs the verdict.  chunked=True enqueues only
        the first FIRST_CHUNK iterations (a tracked frame needs ~5; 50 launch pairs that exit at once would cost more
        host time than the solve): the caller checks `done` and calls _run_rest when the loop is still running.
        Loop state (workspace, status) is double-buffered: the frame begun last and the one begun before it (whose
        verdict may still ask for the rest of its iterations) never share it."""
        cfg = self.cfg
        dev = Xf.device
        idx = self._idx if idx is None else idx
        n = Xk.shape[0]
        if T_rel is None:          # (track_begin passes the one mslam_track_prepare computed)
            T_rel = (T_WCk.inv() * T_WCf).data.reshape(8).contiguous().clone()
        L = _m.lib()
        need = L.mslam_track_workspace_bytes(n)
        cur = getattr(self, "_cur", None)
        sl = self._slots[cur.slot if cur is not None else 0]
        if sl["ws"] is None or sl["ws"].numel() < need:
            sl["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            sl["status"] = torch.zeros(8, dtype=torch.int32, device=dev)
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
                         keep=(T_rel, Xf_c, Xk_c, idx_c, Qk_c, valid_c, K_c), T_rel=T_rel, T_WCk=T_WCk, slot=sl, pose=pose)
        last = min(self.FIRST_CHUNK, int(cfg["max_iters"])) if chunked else int(cfg["max_iters"])
        return self._enqueue(self._job, 0, last)

    def _enqueue(self, job, first, last):
        cfg, sl = self.cfg, job["slot"]
        rc = _m.lib().mslam_track_pose(*job["args"], int(first), int(last), float(cfg["rel_error"]),
                                       float(cfg["delta_norm"]), _m.ptr(sl["status"]), _m.ptr(sl["ws"]),
                                       sl["ws"].numel(), _m.stream_ptr())
        _m.check(rc, "track_pose")
        T_CkCf = Sim3(job["T_rel"].reshape(1, 8))
        return (job["T_WCk"] * T_CkCf if job["pose"] else None), T_CkCf, sl["status"]

    def _run_rest(self):
        """The iterations behind the first chunk (same loop state, same results as one uninterrupted loop)."""
        job = self._cur.job if getattr(self, "_cur", None) is not None and self._cur.job is not None else self._job
        return self._enqueue(job, min(self.FIRST_CHUNK, int(self.cfg["max_iters"])), int(self.cfg["max_iters"]))

    def opt_pose_ray_dist_sim3(self, Xf, Xk, T_WCf, T_WCk, Qk, valid, idx=None):
        """tracker.py:225-266 -> (T_WCf, T_CkCf, ok)."""
        return self._run(False, Xf, Xk, T_WCf, T_WCk, Qk, valid, None, None, idx)

    def opt_pose_calib_sim3(self, Xf, Xk, T_WCf, T_WCk, Qk, valid, meas_k, valid_meas_k, K, img_size, idx=None):
        """tracker.py:268-318 -> (T_WCf, T_CkCf, ok).  meas_k / valid_meas_k are derived from Xk inside
        the kernel (pixel grid + log depth, tracker.py:197-203)."""
        return self._run(True, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx)
