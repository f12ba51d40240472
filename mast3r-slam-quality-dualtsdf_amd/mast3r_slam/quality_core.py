"""Mirror of mast3r_slam/quality_core.py (same function names, arguments and return layouts).  The per-patch
medians and the grid classification run as HIP kernels (csrc/quality.hip); the few scalar / tiny-tensor helpers
stay torch expressions.  The asynchronous service around it (quality_async.py) is scheduling and out of scope."""
import torch
import torch.nn.functional as F

import mslam_hip as _m


def _hw(x, h, w):
    if x is None:
        return None
    if x.ndim == 1 or (x.ndim == 2 and x.shape[-1] == 1):
        x = x.view(h, w)
    return x


def _launch_reduce(x, y, valid, h, w, ps, mode, c_thr=0.0, q_thr=0.0):
    x = x.contiguous().float()
    out = torch.empty((h // ps, w // ps), dtype=torch.float32, device=x.device)
    y = y.contiguous().float() if y is not None else None
    v = valid.to(torch.uint8).contiguous() if valid is not None else None
    rc = _m.lib().mslam_quality_reduce_grid(_m.ptr(x), _m.ptr(y), _m.ptr(v), h, w, ps, mode, float(c_thr), float(q_thr),
                                            _m.ptr(out), _m.stream_ptr())
    _m.check(rc, "quality_reduce_grid")
    return out


def reduce_grid(x, h, w, ps, valid=None, method="median"):
    """quality_core.py:15-29."""
    x = _hw(x, h, w)
    if valid is not None:
        valid = _hw(valid, h, w).to(torch.bool)
    return _launch_reduce(x, None, valid, h, w, ps, 0 if method == "median" else 1)


def upsample_to_hw(g, h, w, mode="bilinear"):
    y = F.interpolate(g[None, None], (h, w), mode=mode, align_corners=False if mode != "nearest" else None)
    return y[0, 0]


def view_weight(t_norm, theta, b0, theta0, device):
    t = torch.clamp(t_norm / b0, 0, 1) if b0 > 0 else torch.ones((), device=device)
    r = torch.clamp(theta / theta0, 0, 1) if theta0 > 0 else torch.ones((), device=device)
    return 0.5 * (t + r)


def ema_delta(prev, inc, alpha):
    new = alpha * prev + (1 - alpha) * inc
    return new, new - prev


def u_from_CQ(C, Q, C_thr, Q_thr, h, w, ps, quant=0.5):
    """quality_core.py:45-52 (uncertainty map and its patch median in one kernel)."""
    return _launch_reduce(_hw(C, h, w), _hw(Q, h, w), None, h, w, ps, 2, C_thr, Q_thr)


def r_from_scalar(r, h, w, ps, valid=None):
    return reduce_grid(r, h, w, ps, valid=valid, method="median")


def valid_grid(valid, h, w, ps):
    v = reduce_grid(valid.float(), h, w, ps, method="mean")
    return (v > 0).float()


def robust_z(x, eps=1e-6):
    m = torch.median(x)
    mad = torch.median(torch.abs(x - m)) + eps
    return (x - m) / mad


def classify(delta_cov, r, u, thr_zr=1.0, thr_zu=1.0, thr_dc=0.02):
    """quality_core.py:66-117 -> (class ids int64, priority f32), shaped like delta_cov."""
    shape = delta_cov.shape
    dc = delta_cov.flatten().contiguous().float()
    rr, uu = r.flatten().contiguous().float(), u.flatten().contiguous().float()
    cls = torch.empty(dc.shape, dtype=torch.long, device=dc.device)
    pri = torch.empty(dc.shape, dtype=torch.float32, device=dc.device)
    rc = _m.lib().mslam_quality_classify(_m.ptr(dc), _m.ptr(rr), _m.ptr(uu), dc.numel(), float(thr_zr), float(thr_zu),
                                         float(thr_dc), _m.ptr(cls), _m.ptr(pri), _m.stream_ptr())
    _m.check(rc, "quality_classify")
    return cls.reshape(shape), pri.reshape(shape)


def pack_result(kfid, dc, r, u, cls, pri, ewma):
    to_numpy = lambda x: x.cpu().numpy() if torch.is_tensor(x) else x
    return {"kf_id": int(kfid), "delta_cov": to_numpy(dc), "r": to_numpy(r), "u": to_numpy(u), "class_id": to_numpy(cls),
            "priority": to_numpy(pri), "cov_ewma": to_numpy(ewma)}


def compute_batch(batch, ps, alpha, b0, theta0, C_thr, Q_thr, thr_zr, thr_zu, thr_dc, device):
    """quality_core.py:122-139."""
    outs = []
    for jb in batch:
        h, w = jb["H"], jb["W"]
        valid = _hw(jb["valid_kf"].to(device), h, w)
        inc = valid_grid(valid, h, w, ps) * view_weight(jb["t_norm"].to(device), jb["theta"].to(device), b0, theta0, device)
        prev = jb.get("cov_ewma", None)
        if prev is None:
            prev = torch.zeros_like(inc, device=device)
        ew, dc = ema_delta(prev, inc, alpha)
        r = r_from_scalar(_hw(jb["r_pix"].to(device), h, w), h, w, ps, valid=valid)
        u = u_from_CQ(jb["Ck"].to(device), jb["Qk"].to(device), C_thr, Q_thr, h, w, ps)
        cls, pri = classify(dc, r, u, thr_zr, thr_zu, thr_dc)
        outs.append(pack_result(jb["kf_id"], dc, r, u, cls, pri, ew))
    return outs
