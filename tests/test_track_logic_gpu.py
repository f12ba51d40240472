"""FrameTracker.track around the pose solver against outcomes recorded from the reference's own class
(tests/golden/track_logic.npz): the inference + matching call and the solver are replaced by the same seeded tensors /
fixed poses on both sides, so this pins the confidence products, validity masks, the match-fraction gate, the
keyframe pointmap fusion through the solved relative pose and the new-keyframe rule."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["normal", "skipped", "new_kf_unique", "solver_fails"])
def test_track_gates_and_fusion(device, golden_dir, monkeypatch, name):
    from lietorch_hip import Sim3
    from mast3r_slam import tracker as T
    from mast3r_slam.config import config
    from mast3r_slam.frame import Frame, KeyframeStore

    fx = np.load(os.path.join(golden_dir, "track_logic.npz"))
    t = lambda k: torch.from_numpy(fx[f"{name}_{k}"]).to(device)
    monkeypatch.setitem(config["tracking"], "filtering_mode", "weighted_pointmap")
    monkeypatch.setitem(config, "use_calib", False)
    for k, v in dict(C_conf=0.0, Q_conf=1.5, min_match_frac=0.05, match_frac_thresh=0.333).items():
        monkeypatch.setitem(config["tracking"], k, v)
    ident = lambda: Sim3.Identity(1, device=device)
    kf = Frame(0, torch.zeros(1, 3, 8, 10, device=device), None, None, None, ident())
    kf.update_pointmap(t("kfX0"), t("kfC0"))
    fr = Frame(1, torch.zeros(1, 3, 8, 10, device=device), None, None, None, ident())
    store = KeyframeStore()
    store.append(kf)
    match = tuple(t(k) for k in ("idx", "vm", "Xff", "Cff", "Qff", "Xkf", "Ckf", "Qkf"))
    monkeypatch.setattr(T, "mast3r_match_asymmetric", lambda model, fi, fj, idx_i2j_init=None: match)
    tr = T.FrameTracker(None, store, device)
    seen = {}
    pose = lambda k: Sim3(torch.from_numpy(fx[k].astype(np.float32)).reshape(1, 8).to(device))

    def solver(use_calib, Xf, Xk, T_WCf, T_WCk, Qk, valid, K, img_size, idx=None, chunked=False, T_rel=None, **_):
        # the enqueue-only form track() uses: (T_WCf, T_CkCf, device status [done, iterations, failed, ...]); track()
        # reaches it even when the match-fraction gate fails (the verdict is read once, after the solve is enqueued)
        seen["Qk"], seen["valid"] = Qk.cpu().numpy(), valid.cpu().numpy()
        status = torch.tensor([1, 3, int(name == "solver_fails"), 0, 0, 0, 0, 0], dtype=torch.int32, device=device)
        return pose("T_new"), pose("T_rel"), status

    monkeypatch.setattr(tr, "_run_async", solver)
    new_kf, info, skipped = tr.track(fr)
    want = fx[f"{name}_ret"]
    assert [bool(new_kf), bool(skipped), tr.idx_f2k is None] == want.tolist()
    if f"{name}_seen_Qk" in fx.files:     # the solver was reached: it saw the same confidence products and mask
        # one ulp: the fixture's sqrt is torch's host kernel (not correctly rounded), the device's is
        np.testing.assert_allclose(seen["Qk"], fx[f"{name}_seen_Qk"], rtol=2e-7, atol=0)
        np.testing.assert_array_equal(seen["valid"], fx[f"{name}_seen_valid"])
    if not skipped:
        for k, v in zip(("Xk", "Ck", "Xf", "Cf", "Qkf", "Qff"), info):
            # keyframe points went through Sim3.act on the device (fp32) vs float64 on the reference side
            np.testing.assert_allclose(v.cpu().numpy(), fx[f"{name}_info_{k}"], rtol=0, atol=2e-6, err_msg=k)
        assert [store[0].N, store[0].N_updates] == fx[f"{name}_kfN"].tolist()
        np.testing.assert_allclose(fr.T_WC.data.cpu().numpy().reshape(-1), fx["T_new"], atol=1e-7)
