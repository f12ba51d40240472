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
