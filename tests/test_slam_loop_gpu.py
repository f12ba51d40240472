"""End-to-end functional test of the hot path WITHOUT the network: a stand-in model hands out the geometry of a
procedural room (pointmaps in the MASt3R two-view convention, position-hashed descriptors, confidences), and the
mirrored SLAM loop of the reference - mono initialisation, FrameTracker.track (asymmetric matching + Sim3 GN +
keyframe decision + pointmap fusion), FactorGraph.add_factors / solve_GN_rays on new keyframes, global TSDF
integration - has to recover the known camera trajectory.  Everything below the model interface is the product
code and the HIP kernels; nothing here comes from the oracle."""
import numpy as np
import pytest
import torch

from mast3r_slam import synthetic

pytestmark = pytest.mark.gpu

H, W = 96, 128


class RoomModel:
    """Same surface as Mast3rHIP (_encode_image, decode_pair): features carry the frame's trajectory index."""

    def __init__(self, device, noise=0.001):
        self.device, self.noise = device, noise

    def _encode_image(self, img, true_shape=None):
        k = int(round(float(img.reshape(-1)[0]) * 1000.0))
        n = (H // 16) * (W // 16)
        feat = torch.full((1, n, 8), float(k), device=self.device)
        return feat, torch.zeros((1, n, 2), dtype=torch.long, device=self.device), None

    def decode_pair(self, feat1, feat2, h, w):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        outs = ([], [])
        for b in range(feat1.shape[0]):
            pr = synthetic.make_pair(int(feat1[b, 0, 0]), int(feat2[b, 0, 0]), h=h, w=w, seed=1, noise=self.noise)
            outs[0].append((pr["X11"], pr["C11"], pr["D11"], pr["Q11"]))
            outs[1].append((pr["X21"], pr["C21"], pr["D21"], pr["Q21"]))
        res = []
        for side in outs:
            X, C, D, Q = (np.stack(v) for v in zip(*side))
            res.append(dict(pts3d=t(X), conf=t(C), desc=t(D), desc_conf=t(Q)))
        return res[0], res[1]


def _rel_pose(T0, Tk):
    """camera k expressed in camera 0 (the gauge of a SLAM run that starts at identity)."""
    inv0 = synthetic.sim3_inv(T0)
    t = synthetic.sim3_act(inv0, Tk[:3][None])[0]
    return t


def test_trajectory_is_recovered(device):
    from lietorch_hip import Sim3
    from mast3r_slam import mast3r_utils as mu
    from mast3r_slam.config import config
    from mast3r_slam.frame import Frame, KeyframeStore
    from mast3r_slam.global_opt import FactorGraph
    from mast3r_slam.tracker import FrameTracker
    from mast3r_slam.tsdf import TSDFGlobalIntegrator, TSDFPoseOptimizer, TSDFVolume

    model = RoomModel(device)
    keyframes = KeyframeStore()
    tracker = FrameTracker(model, keyframes, device)
    graph = FactorGraph(model, keyframes, device=device)
    vol = TSDFVolume(0.03, 0.12, capacity=1 << 18, device=device)
    tcfg = dict(config["tsdf_global"], pre_icp_iters=0)   # fuse only: the reference's TSDF-ICP step is not a contraction
    integ = TSDFGlobalIntegrator(vol, keyframes, tcfg, TSDFPoseOptimizer(vol, keyframes, tcfg, False, device))
    ks = list(range(0, 40, 4))                      # ten frames along the room trajectory
    T0 = synthetic.camera_pose(ks[0])
    last_T = Sim3.Identity(1, device=device)
    est, n_kf_added = {}, 0
    for i, k in enumerate(ks):
        img = torch.full((1, 3, H, W), k / 1000.0, device=device)
        frame = Frame(i, img, torch.tensor([[H, W]]), torch.tensor([[H, W]]), None, Sim3(last_T.data.clone()))
        if i == 0:
            X, C = mu.mast3r_inference_mono(model, frame)
            frame.update_pointmap(X, C)
            keyframes.append(frame)
            est[k] = frame.T_WC.data.reshape(-1).cpu().numpy()
            continue
        new_kf, match_info, skipped = tracker.track(frame)
        assert not skipped and len(match_info) == 6
        last_T = frame.T_WC
        est[k] = frame.T_WC.data.reshape(-1).cpu().numpy()
        if new_kf or i % 3 == 0:                    # the synthetic views overlap a lot: also force a keyframe every third frame
            keyframes.append(frame)
            n_kf_added += 1
            n = len(keyframes)
            assert graph.add_factors([n - 2], [n - 1], config["local_opt"]["min_match_frac"])
            graph.solve_GN_rays()
            integ._integrate_new_keyframes()
    # ---- trajectory error in the gauge of the first camera ---------------------------------------------
    errs = []
    for k in ks[1:]:
        gt_t = _rel_pose(T0, synthetic.camera_pose(k))
        errs.append(np.linalg.norm(est[k][:3] - gt_t))
        assert abs(est[k][7] - 1.0) < 0.02           # metric pointmaps: the Sim3 scale stays 1
    print("translation errors [m]:", np.round(errs, 4))
    assert max(errs) < 0.03, errs                    # the GN stops on the reference's relative-cost rule after ~3 steps
    # keyframe poses after the backend stay on the trajectory as well
    for kf in keyframes._kfs[1:]:
        gt_t = _rel_pose(T0, synthetic.camera_pose(ks[kf.frame_id]))
        assert np.linalg.norm(kf.T_WC.data.reshape(-1)[:3].cpu().numpy() - gt_t) < 0.03
    assert n_kf_added >= 3 and graph.ii.numel() == n_kf_added
    assert vol.stats()["valid_voxels"] > 1000
