"""A calibrated session end to end (BASELINE config 2's code path: config/eval_calib.yaml sets use_calib, main.py:220-230
hands K to the keyframe store): tracking with the pixel + log-depth residual (tracker.py:268-318), the global solve with
gauss_newton_calib (global_opt.py:166-223), pointmaps constrained to the pixel rays (geometry.py) - on the procedural room,
whose renderer uses exactly these intrinsics, with the stand-in model of tests/test_slam_system_gpu.py.  The dataset itself
(TUM fr1_room) and the trained checkpoint are not available offline: this is the harness the real files would run through."""
import numpy as np
import pytest
import torch

from mast3r_slam import synthetic
from test_slam_system_gpu import H, W, RoomModel, _frames, _gauge

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("store_kind,backend", [("list", "inline"), ("shared", "inline"), ("list", "thread")])
def test_calibrated_session_recovers_the_trajectory(device, monkeypatch, store_kind, backend):
    from mast3r_slam.config import config
    from mast3r_slam.frame import KeyframeStore, SharedKeyframes
    from mast3r_slam.slam_system import SlamSystem

    monkeypatch.setitem(config, "use_calib", True)
    monkeypatch.setitem(config["tracking"], "match_frac_thresh", 0.72)
    K = torch.from_numpy(np.ascontiguousarray(synthetic.intrinsics(H, W), np.float32)).to(device)
    store = KeyframeStore() if store_kind == "list" else SharedKeyframes(None, H, W, buffer=16, device=device)
    store.set_intrinsics(K)
    tcfg = dict(config["tsdf_global"], enabled=True, pre_icp_iters=0, max_iterations=0, hash_capacity=1 << 18)
    torch.manual_seed(0)
    system = SlamSystem(RoomModel(device), device, K=K, keyframes=store, frame_group=2, tsdf_global_cfg=tcfg, backend=backend)
    ks = list(range(0, 60, 3))
    frames = _frames(ks, device)
    res = system.run(frames)
    system.shutdown()
    torch.cuda.synchronize()
    n_kf = len(system.keyframes)
    assert n_kf >= 3 and not any(r["try_reloc"] for r in res)
    assert system.factor_graph.ii.numel() == n_kf - 1
    T0 = synthetic.camera_pose(ks[0])
    errs = [np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(k)))
            for f, k in zip(frames[1:], ks[1:])]
    assert max(errs) < 0.05, errs
    for i in range(1, n_kf):
        kf = system.keyframes[i]
        assert kf.K is not None
        err = np.linalg.norm(kf.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(ks[int(kf.frame_id)])))
        assert err < 0.05, (i, err)
    assert system.tsdf_manager.volume.stats()["valid_voxels"] > 1000
