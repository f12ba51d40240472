"""Host-side pieces of the product-loop bench that run without a GPU: the torch formulation of the room renderer
against the numpy room (mast3r_slam/synthetic.py), the pose-proximity retrieval stand-in, and SlamSystem's sizing of the
speculative decode window from the decay of the keyframe rule's value."""
import numpy as np
import torch

from mast3r_slam import synthetic


def test_room_renderer_matches_numpy_room():
    from mast3r_slam.synthetic_gpu import RoomRenderer, camera_pose_t

    H, W = 48, 64
    R = RoomRenderer(torch.device("cpu"), H, W)
    k = torch.tensor([0.0, 3.0, 250.0, 997.0])
    T = camera_pose_t(k)
    for i, kk in enumerate((0, 3, 250, 997)):
        np.testing.assert_allclose(T[i].numpy(), synthetic.camera_pose(kk), atol=1e-12)
        X = R.pointmap(T[i:i + 1])[0].numpy().reshape(H, W, 3)
        np.testing.assert_allclose(X, synthetic.render_pointmap(synthetic.camera_pose(kk), H, W), atol=1e-9)
    a, b = R.pair(torch.tensor([3.0, 12.0]), torch.tensor([0.0, 0.0]), noise=0.0)
    pr = synthetic.make_pair(3, 0, h=H, w=W, noise=0.0)
    np.testing.assert_allclose(a["pts3d"][0].numpy(), pr["X11"], atol=1e-6)
    np.testing.assert_allclose(b["pts3d"][0].numpy(), pr["X21"], atol=1e-6)
    np.testing.assert_allclose(b["desc"][0].numpy(), pr["D21"], atol=1e-6)
    # deterministic per pair, independent of the batch
    a2, b2 = R.pair(torch.tensor([12.0]), torch.tensor([0.0]))
    a1, b1 = R.pair(torch.tensor([3.0, 12.0]), torch.tensor([0.0, 0.0]))
    assert torch.equal(a1["pts3d"][1], a2["pts3d"][0]) and torch.equal(b1["conf"][1], b2["conf"][0])
    img = R.rgb(torch.tensor([9.0]))
    ref = synthetic.render_rgb(synthetic.camera_pose(9), H, W)
    d = np.abs(img[0].numpy() - ref)
    d[0, 0, 0] = 0.0                       # the path-index tag
    assert d.max() < 1e-6 and round(float(img[0, 0, 0, 0]) * 4096) == 9


def test_pose_proximity_retriever():
    from mast3r_slam.synthetic_gpu import PoseProximityRetriever

    class F:
        def __init__(self, fid):
            self.frame_id = fid

    r = PoseProximityRetriever(lambda f: 3 * f.frame_id)
    assert r.update(F(0), add_after_query=True, k=3) == []                 # empty database
    for fid in (8, 16, 24, 32):
        got = r.update(F(fid), add_after_query=True, k=3)
        assert all(0 <= g < len(r.pos) - 1 for g in got) and len(got) <= 3
        assert len(r.pos) - 2 not in got                                    # the consecutive keyframe is never proposed
    n = len(r.pos)
    reloc = r.update(F(33), add_after_query=False, k=2)                    # relocalisation query: nothing is added
    assert len(r.pos) == n and 1 <= len(reloc) <= 2 and n - 1 in reloc     # the nearest view is the most recent one
    far = r.update(F(500), add_after_query=False, k=3)                     # the other side of the path
    assert far == [] or all(np.linalg.norm(r.pos[i] - synthetic.camera_pose(1500)[:3]) < r.max_dist for i in far)


def test_speculative_window_follows_the_keyframe_rule():
    from mast3r_slam.config import config
    from mast3r_slam.slam_system import SlamSystem

    class Tracker:
        last_kf_value = None

    s = SlamSystem.__new__(SlamSystem)
    s.tracker, s._kf_value, s._kf_slope = Tracker(), None, None
    thr = config["tracking"]["match_frac_thresh"]
    assert s._speculative_window(4) == 4                     # nothing known yet (right behind a keyframe change)
    for v in (thr + 0.20, thr + 0.18, thr + 0.16):           # decays 0.02 per frame, far from the threshold
        s.tracker.last_kf_value = v
        s._note_keyframe_rule(False)
    assert s._speculative_window(4) == 4
    for v in (thr + 0.05, thr + 0.03):                        # about one more frame to live
        s.tracker.last_kf_value = v
        s._note_keyframe_rule(False)
    assert s._speculative_window(4) == 1
    s._note_keyframe_rule(True)                               # keyframe changed: full groups again
    assert s._speculative_window(4) == 4
