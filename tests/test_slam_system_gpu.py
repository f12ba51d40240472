"""SlamSystem (the reference's frontend loop + backend in one process, main.py:28-163,325-446) on the
procedural room, WITHOUT the network: a stand-in model hands out the two-view geometry.  Checks the state machine
(INIT -> TRACKING -> RELOC -> TRACKING), the backend hook order, and that frame groups (batched, speculative network
calls) leave every pose bit-identical to one-frame-at-a-time processing."""
import numpy as np
import pytest
import torch

from mast3r_slam import synthetic

pytestmark = pytest.mark.gpu

H, W = 96, 128


class RoomModel:
    """Surface of Mast3rHIP (_encode_image, decode_pair), batch capable; counts its calls and rows."""

    def __init__(self, device, noise=0.001):
        self.device, self.noise = device, noise
        self.enc_calls, self.enc_rows, self.dec_calls, self.dec_rows = 0, 0, 0, 0

    def _encode_image(self, img, true_shape=None):
        self.enc_calls += 1
        B = img.shape[0]
        self.enc_rows += B
        n = (H // 16) * (W // 16)
        k = torch.round(img.reshape(B, -1)[:, 0] * 1000.0)
        feat = k.reshape(B, 1, 1).expand(B, n, 1024).contiguous().float()
        return feat, torch.zeros((B, n, 2), dtype=torch.long, device=self.device), None

    def decode_pair(self, feat1, feat2, h, w):
        self.dec_calls += 1
        self.dec_rows += feat1.shape[0]
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


def _frames(ks, device):
    from mast3r_slam.frame import Frame

    return [Frame(i, torch.full((1, 3, H, W), k / 1000.0, device=device), torch.tensor([[H, W]]), torch.tensor([[H, W]]),
                  torch.zeros(H, W, 3)) for i, k in enumerate(ks)]


def _run(device, ks, frame_group, tsdf=False, backend="inline", shared_store=False, pipeline=False, depth=2, reuse=True,
         encoder_group=None):
    from mast3r_slam.config import config
    from mast3r_slam.slam_system import SlamSystem

    model = RoomModel(device)
    tcfg = dict(config["tsdf_global"], enabled=True, pre_icp_iters=0, max_iterations=0, hash_capacity=1 << 18) if tsdf else None
    torch.manual_seed(0)
    store = None
    if shared_store:   # the reference's slot-buffer layout (frame.py:220-334) instead of the unbounded list
        from mast3r_slam.frame import SharedKeyframes

        store = SharedKeyframes(None, H, W, buffer=16, device=device)
    system = SlamSystem(model, device, frame_group=frame_group, tsdf_global_cfg=tcfg, backend=backend, keyframes=store,
                        pipeline=pipeline, pipeline_depth=depth, encoder_group=encoder_group)
    system.factor_graph.reuse_tracking_decode = reuse
    frames = _frames(ks, device)
    res = system.run(frames)
    system.shutdown()                                   # drains the backend thread, if any
    torch.cuda.synchronize()
    return system, model, frames, res


@pytest.fixture
def eager_keyframes(monkeypatch):
    """The room views overlap a lot (a new keyframe at the reference's 0.333 needs a gap of ~110 trajectory steps):
    raise the threshold so that a gap of ~21 steps (7 frames at stride 3) already gives one."""
    from mast3r_slam.config import config

    monkeypatch.setitem(config["tracking"], "match_frac_thresh", 0.72)


def _gauge(T0, Tk):
    return synthetic.sim3_act(synthetic.sim3_inv(T0), Tk[:3][None])[0]


def test_trajectory_and_backend(device, eager_keyframes):
    from mast3r_slam.frame import Mode

    ks = list(range(0, 60, 3))
    system, model, frames, res = _run(device, ks, 1, tsdf=True)
    assert res[0]["mode"] == Mode.INIT and all(r["mode"] == Mode.TRACKING for r in res[1:])
    n_kf = len(system.keyframes)
    assert n_kf >= 3 and system.stats["keyframes"] == n_kf             # the view changes enough for new keyframes
    assert system.factor_graph.ii.numel() == n_kf - 1                    # one consecutive edge per new keyframe
    T0 = synthetic.camera_pose(ks[0])
    errs = [np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(k)))
            for f, k in zip(frames[1:], ks[1:])]
    assert max(errs) < 0.05, errs
    assert system.tsdf_manager.integrator.next_idx == n_kf               # every keyframe was fused
    assert system.tsdf_manager.volume.stats()["valid_voxels"] > 1000


def test_threaded_backend(device, eager_keyframes):
    """backend="thread": the keyframe tasks run on their own host thread + stream beside tracking (event hand-over of the
    keyframe data between the two streams); interleaving is timing dependent, so the check is on the outcome."""
    ks = list(range(0, 60, 3))
    system, model, frames, res = _run(device, ks, 2, tsdf=True, backend="thread")
    n_kf = len(system.keyframes)
    assert n_kf >= 3 and system.factor_graph.ii.numel() == n_kf - 1
    assert system.tsdf_manager.integrator.next_idx == n_kf
    T0 = synthetic.camera_pose(ks[0])
    errs = [np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(k)))
            for f, k in zip(frames[1:], ks[1:])]
    assert max(errs) < 0.05, errs
    for i in range(1, n_kf):
        kf = system.keyframes[i]
        err = np.linalg.norm(kf.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(ks[kf.frame_id])))
        assert err < 0.05, (i, err)


@pytest.mark.parametrize("group", [2, 4])
def test_frame_groups_are_bit_identical(device, group, eager_keyframes):
    ks = list(range(0, 60, 3))
    s1, m1, f1, r1 = _run(device, ks, 1)
    sg, mg, fg, rg = _run(device, ks, group)
    assert [r["new_kf"] for r in r1] == [r["new_kf"] for r in rg]
    for a, b in zip(f1, fg):
        assert torch.equal(a.T_WC.data, b.T_WC.data)
        assert torch.equal(a.X_canon, b.X_canon) and torch.equal(a.C, b.C)
    for i in range(len(s1.keyframes)):
        assert torch.equal(s1.keyframes[i].T_WC.data, sg.keyframes[i].T_WC.data)
    # fewer, larger network calls; rows decoded in vain only behind keyframe changes
    assert mg.enc_calls <= (len(ks) + group - 1) // group + 1 and m1.enc_calls == len(ks)
    assert mg.dec_calls < m1.dec_calls
    n_kf = len(sg.keyframes)
    assert sg.stats["void_rows"] <= (group - 1) * n_kf
    assert sg.stats["decoded_rows"] == len(ks) - 1 + sg.stats["void_rows"]


def test_larger_encoder_batches_change_nothing(device, eager_keyframes):
    """The look-ahead encoder may run on more frames per call than the decode group (bench default 12 / 6): every frame is
    encoded exactly once, in fewer calls, and nothing downstream changes by a bit (threaded, staged backend)."""
    ks = list(range(0, 66, 3))
    s1, m1, f1, r1 = _run(device, ks, 3, tsdf=True)
    sg, mg, fg, rg = _run(device, ks, 3, tsdf=True, encoder_group=7)
    assert [r["new_kf"] for r in r1] == [r["new_kf"] for r in rg]
    for a, b in zip(f1, fg):
        assert torch.equal(a.T_WC.data, b.T_WC.data) and torch.equal(a.X_canon, b.X_canon) and torch.equal(a.C, b.C)
    for i in range(len(s1.keyframes)):
        assert torch.equal(s1.keyframes[i].T_WC.data, sg.keyframes[i].T_WC.data)
    assert mg.enc_calls < m1.enc_calls and mg.enc_rows == m1.enc_rows == len(ks)


@pytest.mark.parametrize("mode", ["indep_conf", "recent", "first", "weighted_spherical"])
def test_other_filtering_modes(device, mode, eager_keyframes, monkeypatch):
    """The fused fusion launch covers the default 'weighted_pointmap' (frame.py:72-75); every other filtering mode of
    Frame.update_pointmap goes through the op-by-op path of FrameTracker._apply: same loop, frame groups still bit-identical,
    trajectory still recovered."""
    from mast3r_slam.config import config

    monkeypatch.setitem(config["tracking"], "filtering_mode", mode)
    ks = list(range(0, 45, 3))
    s1, m1, f1, r1 = _run(device, ks, 1)
    sg, mg, fg, rg = _run(device, ks, 3)
    assert len(s1.keyframes) >= 2 and [r["new_kf"] for r in r1] == [r["new_kf"] for r in rg]
    for a, b in zip(f1, fg):
        assert torch.equal(a.T_WC.data, b.T_WC.data) and torch.isfinite(a.T_WC.data).all()
    T0 = synthetic.camera_pose(ks[0])
    for i, f in enumerate(f1):
        err = np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(ks[i])))
        assert err < 0.08, (mode, i, err)


def test_shared_keyframe_buffers_give_the_same_trajectory(device, eager_keyframes):
    """SlamSystem on the SharedKeyframes slot buffers (the reference's data layout contract) and on the
    plain list store: identical poses, keyframes and graph."""
    ks = list(range(0, 60, 3))
    sl, _, fl, rl = _run(device, ks, 2)
    ss, _, fs, rs = _run(device, ks, 2, shared_store=True)
    assert len(sl.keyframes) == len(ss.keyframes) >= 3
    for a, b, ra, rb in zip(fl, fs, rl, rs):
        assert torch.equal(ra["pose"], rb["pose"]) and torch.equal(a.T_WC.data, b.T_WC.data)
    for i in range(len(sl.keyframes)):
        ka, kb = sl.keyframes[i], ss.keyframes[i]
        assert ka.frame_id == kb.frame_id and torch.equal(ka.T_WC.data, kb.T_WC.data)
        assert torch.equal(ka.X_canon, kb.X_canon) and torch.equal(ka.C, kb.C) and ka.N == kb.N
    assert torch.equal(sl.factor_graph.ii, ss.factor_graph.ii) and torch.equal(sl.factor_graph.jj, ss.factor_graph.jj)


def test_relocalisation(device):
    """A frame from the other side of the room cannot be tracked (RELOC); the next frame near the map is
    initialised mono, matched against the most recent keyframe and re-enters the graph (main.py:28-71)."""
    from mast3r_slam.frame import Mode

    ks = [0, 3, 6, 9, 250, 12, 15, 18]
    system, model, frames, res = _run(device, ks, 2)
    assert res[4]["try_reloc"] and res[5]["mode"] == Mode.RELOC
    assert system.stats["relocalised"] == 1 and res[6]["mode"] == Mode.TRACKING
    T0 = synthetic.camera_pose(0)
    for f, k in zip(frames[6:], ks[6:]):
        err = np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - _gauge(T0, synthetic.camera_pose(k)))
        assert err < 0.05, (k, err)


def test_threaded_backend_equals_inline_when_drained(device, eager_keyframes):
    """backend="thread" with the backend drained behind every frame must do exactly what the inline order does: the
    solve of a keyframe starts from the previous solve's poses and the TSDF fusions (new keyframe + budgeted
    re-fusions) happen at the poses THAT solve produced, not at the ones copied out in front of it."""
    from mast3r_slam.config import config
    from mast3r_slam.slam_system import SlamSystem

    ks = list(range(0, 60, 3))
    out = {}
    for backend in ("inline", "thread"):
        torch.manual_seed(0)
        tcfg = dict(config["tsdf_global"], enabled=True, pre_icp_iters=0, max_iterations=0, hash_capacity=1 << 18)
        system = SlamSystem(RoomModel(device), device, frame_group=1, tsdf_global_cfg=tcfg, backend=backend)
        frames = _frames(ks, device)
        for i in range(len(frames)):
            system.run(frames, i, i + 1)
            system.drain()
        system.shutdown()
        torch.cuda.synchronize()
        out[backend] = (system, frames, system.tsdf_manager.volume.voxels())
    (si, fi, vi), (st, ft, vt) = out["inline"], out["thread"]
    assert len(si.keyframes) == len(st.keyframes) >= 3
    for a, b in zip(fi, ft):
        assert torch.equal(a.T_WC.data, b.T_WC.data)
    for i in range(len(si.keyframes)):
        assert torch.equal(si.keyframes[i].T_WC.data, st.keyframes[i].T_WC.data), i
    assert np.array_equal(vi[0], vt[0]) and np.array_equal(vi[1], vt[1]) and np.array_equal(vi[2], vt[2])


def test_backlogged_solves_start_from_the_previous_result(device, eager_keyframes):
    """Two keyframe tasks queued back to back while the tracking side has not yet written solve 1's poses into the store:
    solve 2 must start from solve 1's result (taken from the pending commit), and the store ends up with solve 2's."""
    from mast3r_slam.slam_system import SlamSystem

    ks = list(range(0, 60, 3))
    torch.manual_seed(0)
    system = SlamSystem(RoomModel(device), device, frame_group=1, backend="thread")
    frames = _frames(ks, device)
    system.run(frames)
    system.drain()
    n_kf = len(system.keyframes)
    assert n_kf >= 3
    # knock the keyframe poses off so that a solve has something to do
    g = torch.Generator(device=device).manual_seed(5)
    for i in range(1, n_kf):
        kf = system.keyframes[i]
        d = kf.T_WC.data.clone()
        d[..., :3] += 0.02 * torch.randn(3, device=device, generator=g)
        system.keyframes.update_T_WCs(type(kf.T_WC)(d), torch.tensor([i]))
    rec = []
    fg = system.factor_graph
    run_solve = fg.run_solve

    def recording(job):
        before = job["pose_data"].clone()
        run_solve(job)
        rec.append((job["unique_kf_idx_host"].clone(), before, job["pose_data"].clone()))

    fg.run_solve = recording
    apply = system._apply_commits
    system._apply_commits = lambda wait=False: None          # the tracking side is "busy": commits stay pending
    system._queue_backend(n_kf - 1)
    system._queue_backend(n_kf - 1)
    system._worker.drain()
    system._apply_commits = apply
    system.drain()
    torch.cuda.synchronize()
    assert len(rec) == 2 and torch.equal(rec[0][0], rec[1][0])
    pin = 1
    assert (rec[0][1][pin:] - rec[0][2][pin:]).abs().max() > 1e-3          # solve 1 moved the poses ...
    assert torch.equal(rec[1][1][pin:], rec[0][2][pin:])                     # ... and solve 2 started from them
    for r, k in enumerate(rec[1][0].tolist()):
        if r >= pin:
            assert torch.equal(system.keyframes[int(k)].T_WC.data.reshape(8), rec[1][2][r])
    system.shutdown()


@pytest.mark.parametrize("group,ks,depth,gate", [(1, list(range(0, 60, 3)), 2, False), (4, list(range(0, 60, 3)), 3, False),
                                                 (2, [0, 3, 6, 9, 250, 12, 15, 18, 21, 24], 2, False),
                                                 (2, list(range(0, 60, 3)), 1, False), (4, list(range(0, 90, 3)), 2, True)])
def test_pipelined_run_is_bit_identical_to_frame_at_a_time(device, group, ks, depth, gate, eager_keyframes, monkeypatch):
    """SlamSystem.run(pipeline=True) enqueues the matching + solve of up to `depth` frames before it reads frame f's verdict
    and rolls them back (newest first) when f turns out to be a new keyframe / lost; the frame-at-a-time loop (pipeline=False) is the reference
    order.  Every frame's result, pose and pointmap, the keyframes (poses, fused pointmaps, update counts), the graph and
    the voxel table must agree bit for bit - also across a relocalisation (third sequence)."""
    from mast3r_slam.slam_system import SlamSystem

    if not gate:     # speculate blindly: every keyframe change / loss then rolls the frames behind it back
        monkeypatch.setattr(SlamSystem, "_premise_holds", lambda self, in_flight: True)
    sp, mp_, fp, rp = _run(device, ks, group, tsdf=True, pipeline=True, depth=depth)
    ss, ms, fs, rs = _run(device, ks, group, tsdf=True, pipeline=False)
    assert [(r["mode"], r["new_kf"], r["try_reloc"]) for r in rp] == [(r["mode"], r["new_kf"], r["try_reloc"]) for r in rs]
    for a, b, ra, rb in zip(fp, fs, rp, rs):
        assert torch.equal(ra["pose"], rb["pose"]) and torch.equal(a.T_WC.data, b.T_WC.data)
        assert torch.equal(a.X_canon, b.X_canon) and torch.equal(a.C, b.C) and a.N == b.N
    assert len(sp.keyframes) == len(ss.keyframes) >= 2
    for i in range(len(sp.keyframes)):
        ka, kb = sp.keyframes[i], ss.keyframes[i]
        assert ka.frame_id == kb.frame_id and torch.equal(ka.T_WC.data, kb.T_WC.data)
        assert torch.equal(ka.X_canon, kb.X_canon) and torch.equal(ka.C, kb.C) and (ka.N, ka.N_updates) == (kb.N, kb.N_updates)
    fa, fb = sp.factor_graph, ss.factor_graph
    assert torch.equal(fa.ii, fb.ii) and torch.equal(fa.jj, fb.jj) and torch.equal(fa.idx_ii2jj, fb.idx_ii2jj)
    va, vb = sp.tsdf_manager.volume.voxels(), ss.tsdf_manager.volume.voxels()
    assert all(np.array_equal(x, y) for x, y in zip(va, vb))
    if not gate:   # the premise failed where a frame became a keyframe / was lost and other frames stood behind it
        assert sp.stats.get("replayed_frames", 0) >= len(sp.keyframes) - 2
    else:          # with the keyframe-rule prediction as the gate mis-speculation is rare
        assert sp.stats.get("replayed_frames", 0) <= len(sp.keyframes)


def test_pipelined_run_with_solves_that_need_the_second_chunk(device, eager_keyframes, monkeypatch):
    """FIRST_CHUNK = 1: every tracking solve is still running when its verdict is read, so every frame takes the replay
    path (the rest of the iterations, the optimistic effects redone, the frame begun on top rolled back).  Same bits
    as the frame-at-a-time loop, which in turn equals the uninterrupted 8-iteration chunk."""
    from mast3r_slam.tracker import FrameTracker

    ks = list(range(0, 45, 3))
    ref = _run(device, ks, 2, pipeline=False)
    monkeypatch.setattr(FrameTracker, "FIRST_CHUNK", 1)
    got = _run(device, ks, 2, pipeline=True)
    assert got[0].stats.get("replayed_frames", 0) >= len(ks) - 3
    for a, b in zip(ref[2], got[2]):
        assert torch.equal(a.T_WC.data, b.T_WC.data) and torch.equal(a.X_canon, b.X_canon)
    assert len(ref[0].keyframes) == len(got[0].keyframes) >= 2
    for i in range(len(ref[0].keyframes)):
        ka, kb = ref[0].keyframes[i], got[0].keyframes[i]
        assert torch.equal(ka.T_WC.data, kb.T_WC.data) and torch.equal(ka.X_canon, kb.X_canon) and torch.equal(ka.C, kb.C)


@pytest.mark.parametrize("group", [1, 4])
def test_backend_takes_over_the_direction_tracking_decoded(device, group, eager_keyframes):
    """The consecutive edge (previous keyframe, new keyframe) needs decoder(new, previous) - exactly the two-view forward
    tracking ran for the frame that became the new keyframe.  The backend takes that row over instead of decoding it
    again (rows of a batch do not depend on the batch): one decoder + heads row less per keyframe, not a bit changed."""
    ks = list(range(0, 60, 3))
    sa, ma, fa, ra = _run(device, ks, group, tsdf=True, reuse=True)
    sb, mb, fb, rb = _run(device, ks, group, tsdf=True, reuse=False)
    n_kf = len(sa.keyframes)
    assert n_kf == len(sb.keyframes) >= 3
    assert sa.factor_graph.reused_rows == n_kf - 1 and sb.factor_graph.reused_rows == 0
    assert mb.dec_rows - ma.dec_rows == n_kf - 1
    for a, b in zip(fa, fb):
        assert torch.equal(a.T_WC.data, b.T_WC.data)
    for i in range(n_kf):
        assert torch.equal(sa.keyframes[i].T_WC.data, sb.keyframes[i].T_WC.data)
        assert getattr(sa.keyframes[i], "pair_decode", None) is None          # released once the task has used it
    ga, gb = sa.factor_graph, sb.factor_graph
    for k in ("ii", "jj", "idx_ii2jj", "idx_jj2ii", "valid_match_j", "valid_match_i", "Q_ii2jj", "Q_jj2ii"):
        assert torch.equal(getattr(ga, k), getattr(gb, k)), k
