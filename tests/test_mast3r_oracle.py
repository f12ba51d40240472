"""CPU test (-m "not gpu"): the torch-functional MASt3R oracle (oracle/mast3r_ref.py) against outputs
of the reference's own AsymmetricMASt3R classes (tests/golden/mast3r_small.npz: reduced width/depth,
weights = init_state_dict(seed) on both sides, so only inputs/outputs are stored)."""
import os

import numpy as np
import pytest
import torch

from oracle import mast3r_ref as R


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "mast3r_small.npz"))


def test_forward_matches_reference_model(fx):
    c = fx["cfg"]
    cfg = R.Mast3rConfig(enc_dim=int(c[0]), enc_depth=int(c[1]), enc_heads=int(c[2]), dec_dim=int(c[3]),
                         dec_depth=int(c[4]), dec_heads=int(c[5]))
    sd = R.init_state_dict(cfg, seed=int(fx["seed"]))
    img1, img2 = torch.from_numpy(fx["img1"]), torch.from_numpy(fx["img2"])
    H, W = img1.shape[-2:]
    with torch.inference_mode():
        f1, p1 = R.encode_image(sd, cfg, img1)
        f2, p2 = R.encode_image(sd, cfg, img2)
        np.testing.assert_allclose(f1.numpy(), fx["feat1"], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(f2.numpy(), fx["feat2"], atol=2e-5, rtol=1e-4)
        np.testing.assert_array_equal(p1.numpy(), fx["pos1"])
        d1, d2 = R.decoder(sd, cfg, f1, p1, f2, p2)
        assert len(d1) == cfg.dec_depth + 1
        np.testing.assert_allclose(d1[-1].numpy(), fx["dec1_last"], atol=5e-5, rtol=1e-4)
        np.testing.assert_allclose(d2[-1].numpy(), fx["dec2_last"], atol=5e-5, rtol=1e-4)
        np.testing.assert_allclose(d1[6].numpy(), fx["dec1_6"], atol=5e-5, rtol=1e-4)
        for h, toks in ((1, d1), (2, d2)):
            r = R.downstream_head(sd, cfg, h, toks, int(H), int(W))
            for k in ("pts3d", "conf", "desc", "desc_conf"):
                a, b = r[k].numpy(), fx[f"head{h}_{k}"]
                rel = np.linalg.norm(a - b) / np.linalg.norm(b)
                assert rel < 1e-5, (h, k, rel)
                np.testing.assert_allclose(a, b, atol=1e-3, rtol=1e-3, err_msg=f"head{h} {k}")


def test_checkpoint_loader_accepts_upstream_layout(tmp_path):
    """load_mast3r_state_dict reads {'args': Namespace, 'model': state_dict} (mast3r/model.py:24-34) with the
    weights-only loader; a missing file raises FileNotFoundError naming the reference's download step."""
    import argparse

    import pytest

    from mast3r_slam.mast3r_model import load_mast3r_state_dict

    p = tmp_path / "ck.pth"
    torch.save({"args": argparse.Namespace(model="AsymmetricMASt3R(...)"), "model": {"enc_norm.weight": torch.ones(4)}, "epoch": 3}, p)
    sd = load_mast3r_state_dict(str(p))
    assert list(sd) == ["enc_norm.weight"] and torch.equal(sd["enc_norm.weight"], torch.ones(4))
    torch.save({"enc_norm.weight": torch.zeros(2)}, p)     # bare state dict
    assert list(load_mast3r_state_dict(str(p))) == ["enc_norm.weight"]
    with pytest.raises(FileNotFoundError):
        load_mast3r_state_dict(str(tmp_path / "missing.pth"))
