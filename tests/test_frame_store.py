"""SharedKeyframes / SharedStates (frame.py:125-334): buffer shapes and dtypes, view semantics of
__getitem__, copy-in semantics of __setitem__ / append, dirty tracking, pose updates by index tensor, the state
flags.  Runs on the host (the classes are plain torch); the reference module itself needs lietorch and is not
importable here, so this pins the restated contract, not bytes produced by the reference."""
import pytest
import torch

from lietorch_hip import Sim3
from mast3r_slam.config import config
from mast3r_slam.frame import Frame, Mode, SharedKeyframes, SharedStates

H, W = 32, 48


def _frame(i, seed):
    g = torch.Generator().manual_seed(seed)
    f = Frame(i, torch.rand(1, 3, H, W, generator=g), torch.tensor([[H, W]], dtype=torch.int), torch.tensor([[H, W]], dtype=torch.int),
              torch.rand(H, W, 3, generator=g), Sim3(torch.tensor([[0.1 * i, 0.2, 0.3, 0.0, 0.0, 0.0, 1.0, 1.0]])))
    f.X_canon = torch.rand(H * W, 3, generator=g)
    f.C = torch.rand(H * W, 1, generator=g)
    f.feat = torch.rand(1, H * W // 256, 1024, generator=g)
    f.pos = torch.zeros(1, H * W // 256, 2, dtype=torch.long)
    f.N, f.N_updates = 2, 1
    return f


def test_shared_keyframes_contract():
    kfs = SharedKeyframes(None, H, W, buffer=5, device="cpu")
    assert kfs.T_WC.shape == (5, 1, 8) and kfs.X.shape == (5, H * W, 3) and kfs.C.shape == (5, H * W, 1)
    assert kfs.feat.shape == (5, 1, H * W // 256, 1024) and kfs.pos.dtype == torch.long and kfs.uimg.device.type == "cpu"
    assert len(kfs) == 0 and kfs.last_keyframe() is None
    a, b = _frame(7, 0), _frame(9, 1)
    kfs.append(a)
    kfs.append(b)
    assert len(kfs) == 2 and kfs.frame_id_to_index == {7: 0, 9: 1}
    k1 = kfs[1]
    assert k1.frame_id == 9 and k1.N == 2 and k1.N_updates == 1
    assert torch.equal(k1.X_canon, b.X_canon) and torch.equal(k1.img, b.img[0]) and torch.equal(k1.T_WC.data, b.T_WC.data)
    # __getitem__ hands out views: writing through them changes the store; __setitem__ copied, so `b` is untouched
    k1.X_canon[0, 0] = 42.0
    assert kfs.X[1, 0, 0] == 42.0 and b.X_canon[0, 0] != 42.0
    assert kfs.get_dirty_idx().tolist() == [0, 1] and kfs.get_dirty_idx().numel() == 0
    kfs[0] = b
    assert kfs.get_dirty_idx().tolist() == [0] and len(kfs) == 2
    # poses of several keyframes are written with an index tensor (global_opt.py:145-164)
    new = torch.arange(16, dtype=torch.float32).reshape(2, 1, 8)
    kfs.update_T_WCs(Sim3(new), torch.tensor([1, 0]))
    assert torch.equal(kfs.T_WC[1], new[0]) and torch.equal(kfs[0].T_WC.data, new[1])
    assert kfs.last_keyframe().frame_id == 9
    kfs.pop_last()
    assert len(kfs) == 1 and kfs.last_keyframe().frame_id == 9   # slot 0 was overwritten with b above
    for i in range(4):
        kfs.append(_frame(20 + i, i))
    with pytest.raises(IndexError):
        kfs.append(_frame(99, 3))
    old = config["use_calib"]
    try:
        config["use_calib"] = False
        with pytest.raises(AssertionError):
            kfs.set_intrinsics(torch.eye(3))
        config["use_calib"] = True
        kfs.set_intrinsics(torch.eye(3) * 2)
        assert torch.equal(kfs.get_intrinsics(), torch.eye(3) * 2) and kfs[0].K is kfs.K
    finally:
        config["use_calib"] = old


def test_shared_states_contract():
    st = SharedStates(None, H, W, device="cpu")
    assert st.get_mode() == Mode.INIT and not st.is_paused()
    st.set_mode(Mode.TRACKING)
    st.pause()
    assert st.get_mode() == Mode.TRACKING and st.is_paused()
    st.unpause()
    st.dequeue_reloc()
    assert st.reloc_sem.value == 0          # never negative
    st.queue_reloc(); st.queue_reloc(); st.dequeue_reloc()
    assert st.reloc_sem.value == 1
    st.queue_global_optimization(3)
    st.queue_global_optimization(5)
    assert list(st.global_optimizer_tasks) == [3, 5]
    f = _frame(11, 4)
    st.set_frame(f)
    g = st.get_frame()
    assert g.frame_id == 11 and torch.equal(g.X_canon, f.X_canon) and torch.equal(g.img, f.img[0]) and g.X_canon is st.X
    assert torch.equal(g.T_WC.data, f.T_WC.data) and torch.equal(g.feat, f.feat)
