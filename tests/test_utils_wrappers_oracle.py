"""mast3r_utils wrappers (mast3r_utils.py:34-231) against outputs of the reference module itself
(tests/golden/utils_wrappers.npz): stack orders [ii, ji, jj, ij], reshapes, batch splitting of the symmetric
match, the asymmetric match with an initial index, downsample.  A stand-in model and a stand-in matching.match
(the same formulas on both sides, tests/golden/make_golden.py fake_heads / fake_match) make argument order and
tensor routing observable; host tensors, exact."""
import importlib.util
import os

import numpy as np
import pytest
import torch


def _gen(golden_dir):
    spec = importlib.util.spec_from_file_location("make_golden_helpers", os.path.join(golden_dir, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wrappers_match_reference(golden_dir, monkeypatch):
    from mast3r_slam import mast3r_utils as mu
    from mast3r_slam.config import config

    gen = _gen(golden_dir)
    fx = np.load(os.path.join(golden_dir, "utils_wrappers.npz"))
    H, W = 8, 12
    monkeypatch.setattr(mu.matching, "match", gen.fake_match)
    monkeypatch.setitem(config["dataset"], "img_downsample", 1)

    class Model:   # the product's model interface: _encode_image + the fused decode_pair
        def _encode_image(self, img, shape=None):
            return torch.full((1, 4, 3), float(img.reshape(-1)[0])), torch.zeros(1, 4, 2, dtype=torch.long), None

        def decode_pair(self, feat1, feat2, h, w):
            rs = [gen.fake_heads(float(feat1[b].reshape(-1)[0]), float(feat2[b].reshape(-1)[0]), h, w)
                  for b in range(feat1.shape[0])]
            cat = lambda side, k: torch.cat([r[side][k] for r in rs])
            return tuple({k: cat(side, k) for k in ("pts3d", "conf", "desc", "desc_conf")} for side in (0, 1))

    class F:
        def __init__(self, code):
            self.img = torch.full((1, 3, H, W), float(code)); self.img_true_shape = torch.tensor([[H, W]])
            self.feat = None; self.pos = None

    eq = lambda got, key: np.testing.assert_array_equal(got.numpy(), fx[key], err_msg=key)
    model = Model()
    fa, fb, fc = F(1), F(2), F(3)
    X, C = mu.mast3r_inference_mono(model, fa)
    eq(X, "mono_X"); eq(C, "mono_C")
    for k, v in zip("XCDQ", mu.mast3r_symmetric_inference(model, fa, fb)):
        eq(v, f"sym_{k}")
    for k, v in zip("XCDQ", mu.mast3r_asymmetric_inference(model, fb, fc)):
        eq(v, f"asym_{k}")
    feat_i, feat_j = torch.cat((fa.feat, fb.feat)), torch.cat((fb.feat, fc.feat))
    pos = torch.cat((fa.pos, fb.pos))
    shp = [fa.img_true_shape, fb.img_true_shape]
    for k, v in zip("XCDQ", mu.mast3r_decode_symmetric_batch(model, feat_i, pos, feat_j, pos, shp, shp)):
        eq(v, f"batch_{k}")
    res = mu.mast3r_match_symmetric(model, feat_i, pos, feat_j, pos, shp, shp)
    assert len(res) == 8
    for k, v in enumerate(res):
        eq(v, f"msym_{k}")
    init = torch.arange(H * W)[None] % 7
    res = mu.mast3r_match_asymmetric(model, fa, fc, idx_i2j_init=init)
    assert len(res) == 8
    for k, v in enumerate(res):
        eq(v, f"masym_{k}")
    monkeypatch.setitem(config["dataset"], "img_downsample", 2)
    for k, v in zip("XCDQ", mu.mast3r_asymmetric_inference(model, fb, fc)):
        eq(v, f"asym_ds2_{k}")
