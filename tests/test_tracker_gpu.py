"""GPU parity test (-m gpu): the fused tracking GN (csrc/tracker.hip via mast3r_slam.tracker.FrameTracker)
against the numpy restatement of tracker.py (oracle/tracker_py.py).  Tolerance: relative pose 1e-4
(fp32 normal equations summed in a different order, fp64 vs fp32 7x7 solve), same iteration count +-1."""
import numpy as np
import pytest
import torch

import oracle
from oracle import tracker_py
from mast3r_slam import synthetic
from mast3r_slam.config import config

pytestmark = pytest.mark.gpu


def _inputs(h=96, w=128, seed=0):
    g = synthetic.make_graph(n_kf=2, h=h, w=w, seed=seed, pose_noise=0.02, extra_edges=0, stride=6)
    # frame = keyframe 1, keyframe = keyframe 0; directed edge 1 has ii=1 (frame gathered), jj=0
    e = 1
    assert g["ii"][e] > g["jj"][e]
    rng = np.random.default_rng(seed)
    return dict(Xf=g["Xs"][1], Xk=g["Xs"][0], idx=g["idx_ii2jj"][e], valid=g["valid_match"][e, :, 0],
                Qk=rng.uniform(1.0, 4.0, h * w).astype(np.float32), T_WCf=g["Twc"][1], T_WCk=g["Twc"][0],
                K=g["K"], h=h, w=w)


@pytest.mark.parametrize("use_calib", [False, True])
def test_tracking_gn_matches_oracle(device, use_calib):
    from lietorch_hip import Sim3
    from mast3r_slam.tracker import FrameTracker

    d = _inputs()
    cfg = dict(config["tracking"])
    Xf, Xk = d["Xf"], d["Xk"]
    if use_calib:
        # constrain_points_to_ray on both (tracker.py:191-193)
        rays = synthetic.pixel_rays(d["h"], d["w"], d["K"]).reshape(-1, 3).astype(np.float32)
        Xf = (rays * Xf[:, 2:3]).astype(np.float32)
        Xk = (rays * Xk[:, 2:3]).astype(np.float32)
    T_ref, Trel_ref, it_ref = tracker_py.track(use_calib, Xf[d["idx"]], Xk, d["T_WCf"], d["T_WCk"], d["Qk"], d["valid"],
                                               cfg, K=d["K"], img_size=(d["h"], d["w"]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    trk = FrameTracker(None, None, device)
    args = (t(Xf), t(Xk), Sim3(t(d["T_WCf"]).reshape(1, 8)), Sim3(t(d["T_WCk"]).reshape(1, 8)), t(d["Qk"]).reshape(-1, 1),
            t(d["valid"]).reshape(-1, 1))
    if use_calib:
        T_new, T_rel, ok = trk.opt_pose_calib_sim3(*args, None, None, t(d["K"]), (d["h"], d["w"]), idx=t(d["idx"]))
    else:
        T_new, T_rel, ok = trk.opt_pose_ray_dist_sim3(*args, idx=t(d["idx"]))
    assert ok
    assert abs(trk.last_iters - it_ref) <= 1, (trk.last_iters, it_ref)
    np.testing.assert_allclose(T_rel.data.cpu().numpy()[0], Trel_ref, atol=2e-4)
    got = T_new.data.cpu().numpy()[0]
    np.testing.assert_allclose(got[:3], T_ref[:3], atol=3e-4)
    assert min(np.abs(got[3:7] - T_ref[3:7]).max(), np.abs(got[3:7] + T_ref[3:7]).max()) < 3e-4
    # the optimisation must actually have moved toward the ground-truth relative pose
    assert 2 <= trk.last_iters <= 50


def test_tracking_failure_flag(device):
    from lietorch_hip import Sim3
    from mast3r_slam.tracker import FrameTracker

    d = _inputs(24, 32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    trk = FrameTracker(None, None, device)
    none = torch.zeros(24 * 32, 1, dtype=torch.bool, device=device)
    _, _, ok = trk.opt_pose_ray_dist_sim3(t(d["Xf"]), t(d["Xk"]), Sim3(t(d["T_WCf"]).reshape(1, 8)),
                                          Sim3(t(d["T_WCk"]).reshape(1, 8)), t(d["Qk"]).reshape(-1, 1), none,
                                          idx=t(d["idx"]))
    assert not ok  # H = 0 -> Cholesky fails -> the reference's "Cholesky failed" path


@pytest.mark.parametrize("hw", [(48, 64), (384, 512)])
def test_fused_glue_equals_tensor_expressions(device, hw):
    """mslam_track_prepare / _verdict / _fuse against the op-by-op tensor expressions of FrameTracker.track they replace
    (tracker.py:61-75, 147-177; frame.py:72-75 'weighted_pointmap'; lietorch inv / mul / act): bit for bit."""
    import mslam_hip as m
    from lietorch_hip import Sim3

    n = hw[0] * hw[1]
    g = torch.Generator(device="cpu").manual_seed(hw[0])
    r = lambda *s: torch.rand(*s, generator=g)
    idx = torch.randint(0, n, (n,), generator=g).to(device)
    idx[: n // 3] = torch.arange(n // 3, device=device)          # a mix of identity matches and collisions
    vm = (r(n, 1) > 0.2).to(device)
    Qff, Qkf = (1.0 + 3.0 * r(n, 1)).to(device), (1.0 + 3.0 * r(n, 1)).to(device)
    Cf_sum, Ck_sum = (0.5 + 4.0 * r(n, 1)).to(device), (0.5 + 9.0 * r(n, 1)).to(device)
    Nf, Nk, C_conf, Q_conf = 1, 3, 1.3, 2.1
    pose = lambda: Sim3.exp((0.3 * (r(1, 7) - 0.5)).to(device))
    T_WCk, T_WCf = pose(), pose()
    L = m.lib()
    ws = torch.empty(L.mslam_track_prepare_workspace_bytes(n), dtype=torch.uint8, device=device)
    Qk, Ck = torch.empty(n, 1, device=device), torch.empty(n, 1, device=device)
    vo, vk = torch.empty(n, 1, dtype=torch.bool, device=device), torch.empty(n, 1, dtype=torch.bool, device=device)
    T_rel = torch.empty(8, device=device)
    inv = lambda N: float(np.float32(1.0) / np.float32(N))
    m.check(L.mslam_track_prepare(m.ptr(idx), m.ptr(vm), m.ptr(Qff), m.ptr(Qkf), m.ptr(Cf_sum), inv(Nf), m.ptr(Ck_sum), inv(Nk),
                                  C_conf, Q_conf, n, m.ptr(T_WCk.data), m.ptr(T_WCf.data), m.ptr(Qk), m.ptr(Ck), m.ptr(vo),
                                  m.ptr(vk), m.ptr(T_rel), m.ptr(ws), ws.numel(), m.stream_ptr()), "track_prepare")
    # the tensor expressions (as FrameTracker wrote them before the fusion)
    Qk_t = torch.sqrt(Qff[idx] * Qkf)
    Cf_t, Ck_t = (Cf_sum / Nf)[idx], Ck_sum / Nk
    vQ = Qk_t > Q_conf
    vo_t, vk_t = vm & (Cf_t > C_conf) & (Ck_t > C_conf) & vQ, vm & vQ
    hits = torch.zeros(n, dtype=torch.int32, device=device)
    hits.index_add_(0, idx, vm[:, 0].to(torch.int32))
    assert torch.equal(Qk, Qk_t) and torch.equal(Ck, Ck_t) and torch.equal(vo, vo_t) and torch.equal(vk, vk_t)
    assert torch.equal(T_rel, (T_WCk.inv() * T_WCf).data.reshape(8))
    status = torch.tensor([1, 5, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=device)
    v6 = torch.empty(6, device=device)
    m.check(L.mslam_track_verdict(m.ptr(ws), m.ptr(status), n, m.ptr(v6), m.stream_ptr()), "track_verdict")
    want = torch.stack((vo_t.float().mean(), status[1].float(), status[2].float(), vk_t.float().mean(),
                        (hits > 0).float().mean(), status[0].float()))
    assert torch.equal(v6, want), (v6, want)
    assert 0.05 < float(v6[4]) < 0.95
    # fusion of the keyframe's pointmap through the solved relative pose
    Xkf, Ckf = (r(n, 3) * 4 - 2).to(device), (0.5 + 2.0 * r(n, 1)).to(device)
    Xc = (r(n, 3) * 4 - 2).to(device)
    T_CkCf = Sim3(T_rel.reshape(1, 8))
    T_out, X_new, C_new = torch.empty(1, 8, device=device), torch.empty_like(Xc), torch.empty_like(Ck_sum)
    m.check(L.mslam_track_fuse(m.ptr(T_WCk.data), m.ptr(T_rel), m.ptr(Xkf), m.ptr(Ckf), m.ptr(Xc), m.ptr(Ck_sum), n,
                               m.ptr(T_out), m.ptr(X_new), m.ptr(C_new), m.stream_ptr()), "track_fuse")
    Xkk = T_CkCf.act(Xkf)
    assert torch.equal(T_out, (T_WCk * T_CkCf).data.reshape(1, 8))
    assert torch.equal(X_new, ((Ck_sum * Xc) + (Ckf * Xkk)) / (Ck_sum + Ckf))
    assert torch.equal(C_new, Ck_sum + Ckf)


def test_fused_glue_edge_cases(device):
    """mslam_track_prepare / _verdict at a point count that is no multiple of a wave, a block or 16 bytes, with no valid match
    at all and with every match pointing at one pixel."""
    import mslam_hip as m
    from lietorch_hip import Sim3

    L = m.lib()
    for n, mode in ((1003, "none"), (1003, "one"), (257, "all"), (5, "all")):
        g = torch.Generator().manual_seed(n)
        idx = torch.randint(0, n, (n,), generator=g).to(device)
        vm = torch.ones(n, 1, dtype=torch.bool, device=device)
        if mode == "none":
            vm.zero_()
        if mode == "one":
            idx.fill_(n - 1)
        Qff = Qkf = torch.full((n, 1), 4.0, device=device)
        Cs = torch.full((n, 1), 2.0, device=device)
        T = Sim3.Identity(1, device=device)
        ws = torch.empty(L.mslam_track_prepare_workspace_bytes(n), dtype=torch.uint8, device=device)
        ws.fill_(255)                                   # the call must not rely on a zeroed workspace
        Qk, Ck = torch.empty(n, 1, device=device), torch.empty(n, 1, device=device)
        vo, vk = torch.empty(n, 1, dtype=torch.bool, device=device), torch.empty(n, 1, dtype=torch.bool, device=device)
        T_rel = torch.empty(8, device=device)
        m.check(L.mslam_track_prepare(m.ptr(idx), m.ptr(vm), m.ptr(Qff), m.ptr(Qkf), m.ptr(Cs), 1.0, m.ptr(Cs), 1.0, 0.0, 1.5, n,
                                      m.ptr(T.data), m.ptr(T.data), m.ptr(Qk), m.ptr(Ck), m.ptr(vo), m.ptr(vk), m.ptr(T_rel),
                                      m.ptr(ws), ws.numel(), m.stream_ptr()), "track_prepare")
        status = torch.tensor([0, 2, 1, 0, 0, 0, 0, 0], dtype=torch.int32, device=device)
        v6 = torch.empty(6, device=device)
        m.check(L.mslam_track_verdict(m.ptr(ws), m.ptr(status), n, m.ptr(v6), m.stream_ptr()), "track_verdict")
        hits = torch.zeros(n, dtype=torch.int32, device=device)
        hits.index_add_(0, idx, vm[:, 0].to(torch.int32))
        want = torch.stack((vm.float().mean(), status[1].float(), status[2].float(), vm.float().mean(),
                            (hits > 0).float().mean(), status[0].float()))
        assert torch.equal(v6, want), (n, mode, v6, want)
        assert torch.equal(vo, vm) and torch.equal(vk, vm) and torch.equal(Qk, torch.full_like(Qk, 4.0))
        assert torch.equal(T_rel, T.data.reshape(8))
