"""Output writers (mast3r_slam/evaluate.py) against the reference module's own output on the same three
keyframes (tests/golden/evaluate.npz: the TUM trajectory text and the vertex array the reference hands to plyfile)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _read_ply(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
    n = int(lines[2].split()[2])
    assert lines[2].split()[:2] == ["element", "vertex"]
    back = {"float": "<f4", "uchar": "u1"}
    dt = np.dtype([(ln.split()[2], back[ln.split()[1]]) for ln in lines[3:]])
    assert len(body) == n * dt.itemsize
    return np.frombuffer(body, dtype=dt)


def _keyframes(fx, device):
    from lietorch_hip import Sim3
    from mast3r_slam.frame import Frame, KeyframeStore

    store = KeyframeStore()
    for i, fid in enumerate(fx["frame_ids"]):
        H, W = fx[f"uimg_{i}"].shape[:2]
        kf = Frame(int(fid), torch.zeros(1, 3, H, W, device=device), torch.tensor([[H, W]]), torch.tensor([[H, W]]),
                   torch.from_numpy(fx[f"uimg_{i}"]), Sim3(torch.from_numpy(fx[f"T_{i}"]).reshape(1, 8).to(device)),
                   torch.from_numpy(fx[f"X_{i}"]).to(device), torch.from_numpy(fx[f"C_{i}"]).to(device))
        kf.N = 1
        store.append(kf)
    return store


def test_save_traj_and_reconstruction(device, golden_dir, tmp_path):
    from mast3r_slam import evaluate as ev

    fx = np.load(os.path.join(golden_dir, "evaluate.npz"))
    store = _keyframes(fx, device)
    ev.save_traj(tmp_path, "traj.txt", list(fx["timestamps"]), store)
    got, want = open(tmp_path / "traj.txt").read().splitlines(), str(fx["traj_txt"]).splitlines()
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert g.split()[0] == w.split()[0]                                   # timestamp text
        np.testing.assert_allclose([float(v) for v in g.split()[1:]], [float(v) for v in w.split()[1:]], rtol=0, atol=1e-7)
    ev.save_reconstruction(tmp_path, "rec.ply", store, float(fx["c_conf_threshold"]))
    pcd = _read_ply(tmp_path / "rec.ply")
    assert list(pcd.dtype.names) == list(fx["ply_names"]) and not bool(fx["ply_text"])
    assert [pcd.dtype[n].str.lstrip("<|") for n in pcd.dtype.names] == [t.lstrip("<|") for t in fx["ply_types"]]
    assert len(pcd) == len(fx["ply_x"])
    for c in ("red", "green", "blue"):
        np.testing.assert_array_equal(pcd[c], fx["ply_" + c])
    for c in "xyz":
        np.testing.assert_allclose(pcd[c], fx["ply_" + c], rtol=0, atol=2e-6)


def test_save_ply_with_quality_and_keyframes(device, golden_dir, tmp_path):
    import PIL.Image

    from mast3r_slam import evaluate as ev

    fx = np.load(os.path.join(golden_dir, "evaluate.npz"))
    store = _keyframes(fx, device)
    H, W = fx["uimg_0"].shape[:2]
    gh, gw = 3, 4
    grids = {int(fid): dict(r=torch.rand(gh, gw), delta_cov=np.random.rand(gh, gw), u=torch.rand(gh, gw),
                            class_id=torch.randint(0, 4, (gh, gw)), priority=torch.rand(gh, gw)) for fid in fx["frame_ids"][:2]}
    ev.save_ply_with_quality(tmp_path, "q.ply", store, float(fx["c_conf_threshold"]), grids, patch_size=8)
    pcd = _read_ply(tmp_path / "q.ply")
    assert list(pcd.dtype.names) == ["x", "y", "z", "red", "green", "blue", "r", "delta_cov", "u", "class_id", "priority"]
    assert len(pcd) == len(fx["ply_x"])
    np.testing.assert_allclose(pcd["x"], fx["ply_x"], rtol=0, atol=2e-6)
    n_last = int((fx["C_2"].reshape(-1) > float(fx["c_conf_threshold"])).sum())
    assert np.all(pcd["r"][-n_last:] == 0) and np.all(pcd["class_id"][-n_last:] == 0)   # no quality result for the third keyframe
    assert pcd["r"][:-n_last].min() >= 0 and pcd["r"][:-n_last].max() <= 1 and pcd["class_id"].max() <= 3
    # upsampling rules: a constant grid stays constant, nearest picks source cells
    g = np.arange(12, dtype=np.float32).reshape(3, 4)
    np.testing.assert_array_equal(ev._resize_grid(g, 6, 8, "nearest"), np.repeat(np.repeat(g, 2, 0), 2, 1))
    np.testing.assert_allclose(ev._resize_grid(np.full((3, 4), 2.5, np.float32), H, W, "linear"), 2.5)
    lin = ev._resize_grid(g, 6, 8, "linear")
    assert lin[0, 0] == 0 and lin[-1, -1] == 11 and abs(lin[0, 1] - 0.25) < 1e-6
    ev.save_keyframes(tmp_path / "kf", list(fx["timestamps"]), store)
    t0 = fx["timestamps"][int(fx["frame_ids"][0])]
    img = np.asarray(PIL.Image.open(tmp_path / "kf" / f"{t0}.png"))
    np.testing.assert_array_equal(img, (fx["uimg_0"] * 255).astype(np.uint8))
