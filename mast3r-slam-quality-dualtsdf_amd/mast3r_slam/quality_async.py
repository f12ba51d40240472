"""Mirror of mast3r_slam/quality_async.py (AsynchronousQualityService, lines 48-310): the same surface
(`submit / poll / get / get_by_kf_id / get_by_frame_id / register_callback / shutdown`, the two caches, the persisted
coverage EWMA per keyframe, the sliding-window global statistics) without the worker thread: `poll()` computes the
queued jobs itself, in batches of `quality.batch_size`, with quality_core.compute_batch (csrc/quality.hip kernels).
Jobs may carry device tensors (the reference's tracker converts to numpy and the worker converts back, tracker.py:128-140).

Reference quirk kept: TSDFRefiner._schedule_refinement asks `get(frame_id)` (tsdf_refine.py:357-358) while `get`
looks up by KEYFRAME INDEX (quality_async.py:196-199), so a result is found only where the two numbers coincide;
`lookup_both=True` makes `get` fall back to the frame-id cache (not the reference's behaviour; the bench says so)."""
import threading
from collections import deque

import numpy as np
import torch

from mast3r_slam.config import config
from mast3r_slam.quality_core import compute_batch


def _prep_job(j, dev):
    """quality_async.py:12-46.  A job may carry its residual map as a deferred expression (`lazy`, FrameTracker): it is
    evaluated here, i.e. only for the jobs that are actually computed."""
    ev = j.get("event")
    if ev is not None:                           # the job's tensors were produced on the submitting thread's stream
        torch.cuda.current_stream().wait_event(ev)
    if "lazy" in j:
        j = dict(j)
        if torch.cuda.is_available():
            for t in j.get("keep", ()):          # inputs allocated on the tracking stream, read on this one
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(torch.cuda.current_stream(t.device))
        j.update(j.pop("lazy")())
    out = {"kf_id": int(j["kf_id"]), "frame_id": int(j.get("frame_id", j["kf_id"])), "H": int(j["H"]), "W": int(j["W"]),
           "t_norm": torch.as_tensor(j["t_norm"], device=dev, dtype=torch.float32),
           "theta": torch.as_tensor(j["theta"], device=dev, dtype=torch.float32)}

    def tens(x, dtype=torch.float32):
        if x is None:
            return None
        t = torch.from_numpy(x) if isinstance(x, np.ndarray) else (x if torch.is_tensor(x) else torch.as_tensor(x))
        return t.to(dev) if t.dtype == torch.bool else t.to(device=dev, dtype=dtype)

    out["valid_kf"] = tens(j["valid_kf"], torch.bool)
    out["r_pix"] = tens(j["r_pix"]) if j.get("r_pix") is not None else torch.zeros(out["H"] * out["W"], device=dev)
    out["Ck"], out["Qk"] = tens(j["Ck"]), tens(j["Qk"])
    ew = j.get("cov_ewma", None)
    out["cov_ewma"] = tens(ew) if ew is not None else None
    return out


class SynchronousQualityService:
    def __init__(self, manager=None, device=None, lookup_both=False, max_jobs=100):
        self.jobs = deque(maxlen=max_jobs)          # job_q (maxsize 100; a full queue drops the NEW job there, the oldest here)
        self.cache_by_kf_id, self.cache_by_frame_id = {}, {}
        self.ewma_state = {}
        self.callbacks, self.callback_lock = [], threading.Lock()
        self.global_stats = {"r_median": 1.0, "r_mad": 0.5, "u_median": 0.5, "u_mad": 0.2}
        self.stats_window = deque(maxlen=50)
        self.lookup_both = lookup_both
        self.lock = threading.Lock()
        qcfg = config.get("quality", {})
        cov = qcfg.get("metrics", {}).get("coverage", {})
        thr = qcfg.get("thresholds", {})
        self.cfg = {"patch_size": int(qcfg.get("patch_size", 16)), "batch_size": int(qcfg.get("batch_size", 4)),
                    "alpha": float(cov.get("alpha_ema", 0.8)), "b0": float(cov.get("b0", 0.15)),
                    "theta0": float(cov.get("theta0_deg", 10.0)) * (3.1415926535 / 180.0),
                    "C_thr": float(config.get("tracking", {}).get("C_conf", 0.0)),
                    "Q_thr": float(config.get("tracking", {}).get("Q_conf", 0.0)),
                    "tzr": float(thr.get("z_r", 1.0)), "tzu": float(thr.get("z_u", 1.0)), "tdc": float(thr.get("d_cov", 0.02))}
        self.device = device or "cuda"

    def submit(self, job):
        """quality_async.py:107-117: the keyframe's persisted coverage EWMA rides along.  A newer job for the same
        keyframe replaces a queued one (the result cache is keyed by keyframe anyway)."""
        if torch.cuda.is_available() and any(torch.is_tensor(v) and v.is_cuda for v in job.values()):
            job["event"] = torch.cuda.Event()
            job["event"].record()
        with self.lock:
            kf_id = job.get("kf_id")
            if kf_id is not None and kf_id in self.ewma_state:
                job["cov_ewma"] = self.ewma_state[kf_id]
            for k, old in enumerate(self.jobs):
                if old.get("kf_id") == kf_id:
                    self.jobs[k] = job
                    return
            self.jobs.append(job)

    def register_callback(self, callback):
        with self.callback_lock:
            self.callbacks.append(callback)

    def poll(self):
        """Computes every queued job (the reference's worker loop body, quality_async.py:211-250) and files the results
        (_process_result, :137-161).  Returns the number of results."""
        with self.lock:
            jobs, n = list(self.jobs), 0
            self.jobs.clear()
        c = self.cfg
        for b0 in range(0, len(jobs), c["batch_size"]):
            batch = [_prep_job(j, self.device) for j in jobs[b0:b0 + c["batch_size"]]]
            results = compute_batch(batch, ps=c["patch_size"], alpha=c["alpha"], b0=c["b0"], theta0=c["theta0"],
                                    C_thr=c["C_thr"], Q_thr=c["Q_thr"], thr_zr=c["tzr"], thr_zu=c["tzu"], thr_dc=c["tdc"],
                                    device=self.device)
            for job, msg in zip(batch, results):
                msg["frame_id"] = job["frame_id"]
                self._process_result(msg)
                n += 1
        return n

    def _process_result(self, msg):
        kf_id, frame_id = msg.get("kf_id"), msg.get("frame_id")
        if kf_id is not None:
            self.cache_by_kf_id[kf_id] = msg
            if "cov_ewma" in msg:
                self.ewma_state[kf_id] = msg["cov_ewma"]
        if frame_id is not None:
            self.cache_by_frame_id[frame_id] = msg
        self.stats_window.append({"r": msg.get("r"), "u": msg.get("u")})
        if len(self.stats_window) >= 10:   # quality_async.py:163-194
            for key in ("r", "u"):
                vals = [np.asarray(x[key]).ravel() for x in self.stats_window if x[key] is not None]
                if vals:
                    v = np.concatenate(vals).astype(np.float32)
                    med = float(np.median(v))
                    self.global_stats[key + "_median"] = med
                    self.global_stats[key + "_mad"] = float(np.median(np.abs(v - med)))
        with self.callback_lock:
            for cb in self.callbacks:
                cb(msg)

    def get(self, kf_id):
        self.poll()
        r = self.cache_by_kf_id.get(int(kf_id), None)
        if r is None and self.lookup_both:
            r = self.cache_by_frame_id.get(int(kf_id), None)
        return r

    def get_by_kf_id(self, kf_id):
        self.poll()
        return self.cache_by_kf_id.get(int(kf_id), None)

    def get_by_frame_id(self, frame_id):
        self.poll()
        return self.cache_by_frame_id.get(int(frame_id), None)

    def shutdown(self, timeout=1.0):
        self.poll()
