"""GPU parity tests (-m gpu) for the MASt3R forward (csrc/gemm.hip, attention.hip, mast3r.hip) through the
C ABI, against the torch-fp32 oracle (oracle/mast3r_ref.py, itself pinned to the reference model classes
by tests/golden/mast3r_small.npz) and against that fixture directly.

Stated tolerances (bf16 MFMA operands, fp32 accumulate):
  encoder / decoder tokens   rel-L2 <= 2e-2
  pts3d                      rel-L2 <= 3e-2 (after expm1)
  conf, desc_conf            rel-L2 <= 5e-2 (after exp)
  desc                       mean cosine >= 0.999
"""
import os

import numpy as np
import pytest
import torch

from oracle import mast3r_ref as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _check_heads(res, ref):
    assert _rel(res["pts3d"].cpu().numpy(), ref["pts3d"]) <= 3e-2
    assert _rel(res["conf"].cpu().numpy(), ref["conf"]) <= 5e-2
    assert _rel(res["desc_conf"].cpu().numpy(), ref["desc_conf"]) <= 5e-2
    d = res["desc"].cpu().numpy()
    cos = (d * np.asarray(ref["desc"])).sum(-1)
    assert cos.mean() >= 0.999, cos.mean()
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-4)


def _model(cfg, seed, device):
    from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP

    sd = R.init_state_dict(cfg, seed=seed)
    hip_cfg = Mast3rConfig(cfg.enc_dim, cfg.enc_depth, cfg.enc_heads, cfg.dec_dim, cfg.dec_depth, cfg.dec_heads)
    return sd, Mast3rHIP(sd, hip_cfg, device=device)


def test_small_model_matches_reference_fixture(device, golden_dir):
    fx = np.load(os.path.join(golden_dir, "mast3r_small.npz"))
    c = fx["cfg"]
    cfg = R.Mast3rConfig(enc_dim=int(c[0]), enc_depth=int(c[1]), enc_heads=int(c[2]), dec_dim=int(c[3]),
                         dec_depth=int(c[4]), dec_heads=int(c[5]))
    sd, model = _model(cfg, int(fx["seed"]), device)
    img1, img2 = torch.from_numpy(fx["img1"]).to(device), torch.from_numpy(fx["img2"]).to(device)
    H, W = img1.shape[-2:]
    ts = torch.tensor([[H, W]])
    f1, p1, _ = model._encode_image(img1, ts)
    f2, p2, _ = model._encode_image(img2, ts)
    np.testing.assert_array_equal(p1.cpu().numpy(), fx["pos1"])
    assert _rel(f1.cpu().numpy(), fx["feat1"]) <= 2e-2, _rel(f1.cpu().numpy(), fx["feat1"])
    assert _rel(f2.cpu().numpy(), fx["feat2"]) <= 2e-2
    # feed the REFERENCE encoder tokens into the decoder so the decoder/head check is independent
    r1, r2, d1, d2 = model.decode_pair(torch.from_numpy(fx["feat1"]).to(device), torch.from_numpy(fx["feat2"]).to(device),
                                       H, W, return_tokens=True)
    assert _rel(d1.cpu().numpy(), fx["dec1_last"]) <= 2e-2, _rel(d1.cpu().numpy(), fx["dec1_last"])
    assert _rel(d2.cpu().numpy(), fx["dec2_last"]) <= 2e-2
    for h, r in ((1, r1), (2, r2)):
        _check_heads(r, {k: fx[f"head{h}_{k}"] for k in ("pts3d", "conf", "desc", "desc_conf")})
    # the reference call sequence (mast3r_utils.py:36-39) works on the model object
    dec1, dec2 = model._decoder(f1, p1, f2, p2)
    assert len(dec1) == cfg.dec_depth + 1
    res1 = model._downstream_head(1, [t.float() for t in dec1], ts)
    res2 = model._downstream_head(2, [t.float() for t in dec2], ts)
    assert res1["pts3d"].shape == (1, H, W, 3) and res2["desc"].shape == (1, H, W, 24)


@pytest.mark.parametrize("shape", [(2, 128, 160), (1, 384, 512)])
def test_medium_model_matches_oracle(device, shape):
    """Wider model (dims 256/192... head_dim 64), batch > 1 and the full 512x384 token grid (24x32):
    exercises the 128x128 GEMM tiles, 12 K/V tiles in attention and every conv shape of the DPT."""
    B, H, W = shape
    cfg = R.Mast3rConfig(enc_dim=256, enc_depth=2, enc_heads=4, dec_dim=192, dec_depth=12, dec_heads=3)
    sd, model = _model(cfg, 99, device)
    g = torch.Generator().manual_seed(3)
    img1 = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    img2 = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    with torch.inference_mode():
        f1, p1 = R.encode_image(sd, cfg, img1)
        f2, p2 = R.encode_image(sd, cfg, img2)
        d1, d2 = R.decoder(sd, cfg, f1, p1, f2, p2)
        ref1 = R.downstream_head(sd, cfg, 1, d1, H, W)
        ref2 = R.downstream_head(sd, cfg, 2, d2, H, W)
    hf1, _, _ = model._encode_image(img1.to(device))
    hf2, _, _ = model._encode_image(img2.to(device))
    assert _rel(hf1.cpu().numpy(), f1.numpy()) <= 2e-2, _rel(hf1.cpu().numpy(), f1.numpy())
    assert _rel(hf2.cpu().numpy(), f2.numpy()) <= 2e-2
    r1, r2, t1, t2 = model.decode_pair(f1.to(device), f2.to(device), H, W, return_tokens=True)
    assert _rel(t1.cpu().numpy(), d1[-1].numpy()) <= 2e-2, _rel(t1.cpu().numpy(), d1[-1].numpy())
    assert _rel(t2.cpu().numpy(), d2[-1].numpy()) <= 2e-2
    _check_heads(r1, {k: v.numpy() for k, v in ref1.items()})
    _check_heads(r2, {k: v.numpy() for k, v in ref2.items()})


def test_mast3r_utils_wrappers(device):
    """mast3r_inference_mono / mast3r_match_asymmetric / mast3r_match_symmetric return the reference's
    tuple layouts (mast3r_utils.py:118-231)."""
    from mast3r_slam import mast3r_utils as mu
    from mast3r_slam.frame import Frame

    cfg = R.Mast3rConfig(enc_dim=128, enc_depth=1, enc_heads=2, dec_dim=128, dec_depth=12, dec_heads=2)
    sd, model = _model(cfg, 5, device)
    H, W = 64, 96
    g = torch.Generator().manual_seed(0)
    mk = lambda i: Frame(i, (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(device), torch.tensor([[H, W]]),
                         torch.tensor([[H, W]]), None)
    fa, fb = mk(0), mk(1)
    Xii, Cii = mu.mast3r_inference_mono(model, fa)
    assert Xii.shape == (H * W, 3) and Cii.shape == (H * W, 1) and fa.feat is not None
    out = mu.mast3r_match_asymmetric(model, fb, fa)
    idx, valid, Xff, Cff, Qff, Xkf, Ckf, Qkf = out
    assert idx.shape == (1, H * W) and idx.dtype == torch.int64 and valid.shape == (1, H * W, 1)
    assert Xff.shape == (H * W, 3) and Qkf.shape == (H * W, 1)
    feat_i = torch.cat([fa.feat, fb.feat]); feat_j = torch.cat([fb.feat, fa.feat])
    pos = torch.cat([fa.pos, fb.pos])
    shp = [fa.img_true_shape, fb.img_true_shape]
    res = mu.mast3r_match_symmetric(model, feat_i, pos, feat_j, pos, shp, shp)
    assert len(res) == 8 and res[0].shape == (2, H * W) and res[4].shape == (2, H * W, 1)
    # batched symmetric decode == per-edge decode (the reference's python loop), bitwise
    X, C, D, Q = mu.mast3r_decode_symmetric_batch(model, feat_i, pos, feat_j, pos, shp, shp)
    X1, _, _, _ = mu.mast3r_decode_symmetric_batch(model, feat_i[:1], pos[:1], feat_j[:1], pos[:1], shp[:1], shp[:1])
    assert torch.equal(X[:, :1], X1)


def test_frame_group_matches_per_frame(device):
    """encode_frames / mast3r_asymmetric_inference_group: a group of frames through ONE encoder and ONE decoder call
    gives bit-identical per-frame results to the reference's one-frame-at-a-time calls; a stashed result is used
    once, and only for the keyframe it was decoded against."""
    from mast3r_slam import mast3r_utils as mu
    from mast3r_slam.frame import Frame

    cfg = R.Mast3rConfig(enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=12, dec_heads=2)
    sd, model = _model(cfg, 9, device)
    H, W = 64, 96
    g = torch.Generator().manual_seed(3)
    imgs = [(torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(device) for _ in range(5)]
    mk = lambda i: Frame(i, imgs[i], torch.tensor([[H, W]]), torch.tensor([[H, W]]), None)
    kf, kf2 = mk(0), mk(4)
    single = []
    for i in (1, 2, 3):
        f = mk(i)
        single.append((mu.mast3r_asymmetric_inference(model, f, kf), f.feat.clone()))
    group = [mk(i) for i in (1, 2, 3)]
    kfg = mk(0)
    mu.mast3r_asymmetric_inference_group(model, group, kfg)
    calls = []
    real = model.decode_pair
    model.decode_pair = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    for f, (ref, feat) in zip(group, single):
        assert torch.equal(f.feat, feat)
        out = mu.mast3r_asymmetric_inference(model, f, kfg)
        assert f.decoded is None
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    assert not calls                               # all three came from the group call
    mu.mast3r_asymmetric_inference(model, group[0], kfg)
    assert len(calls) == 1                         # the stash is used once
    mu.mast3r_asymmetric_inference_group(model, group[1:], kfg)
    out = mu.mast3r_asymmetric_inference(model, group[1], kf2)   # the keyframe changed: recomputed against the new one
    assert len(calls) == 3
    ref = mu.mast3r_asymmetric_inference(model, mk(2), mk(4))
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    model.decode_pair = real


def test_graph_replay_matches_eager(device):
    """use_graphs=True replays captured HIP graphs (two-stream decoder included); results are bit-identical
    to the eager launches and stay valid after later calls (fresh output tensors)."""
    from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP

    cfg = R.Mast3rConfig(enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=12, dec_heads=2)
    sd = R.init_state_dict(cfg, seed=11)
    hc = Mast3rConfig(cfg.enc_dim, cfg.enc_depth, cfg.enc_heads, cfg.dec_dim, cfg.dec_depth, cfg.dec_heads)
    eager, graphed = Mast3rHIP(sd, hc, device=device), Mast3rHIP(sd, hc, device=device, use_graphs=True)
    g = torch.Generator().manual_seed(5)
    imgs = [(torch.rand(1, 3, 64, 96, generator=g) * 2 - 1).to(device) for _ in range(3)]
    kept = []
    for k in range(3):   # call 0 captures, calls 1-2 replay
        fe = eager._encode_image(imgs[k])[0]
        fg = graphed._encode_image(imgs[k])[0]
        assert torch.equal(fe, fg)
        re1, re2 = eager.decode_pair(fe, eager._encode_image(imgs[0])[0], 64, 96)
        rg1, rg2 = graphed.decode_pair(fg, graphed._encode_image(imgs[0])[0], 64, 96)
        for key in ("pts3d", "conf", "desc", "desc_conf"):
            assert torch.equal(re1[key], rg1[key]) and torch.equal(re2[key], rg2[key]), key
        kept.append((rg1["pts3d"], re1["pts3d"].clone()))
    for got, want in kept:   # earlier results were not overwritten by later replays
        assert torch.equal(got, want)


def test_concurrent_calls_from_two_threads(device):
    """The SLAM frontend and backend call the model at the same time from different host threads and streams
    (bench.py): every (call kind, stream) has its own arena and fork context, so concurrent results are
    bit-identical to the same calls issued one after the other."""
    import threading

    from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP

    cfg = R.Mast3rConfig(enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=12, dec_heads=2)
    sd = R.init_state_dict(cfg, seed=21)
    model = Mast3rHIP(sd, Mast3rConfig(cfg.enc_dim, cfg.enc_depth, cfg.enc_heads, cfg.dec_dim, cfg.dec_depth, cfg.dec_heads),
                      device=device)
    g = torch.Generator().manual_seed(9)
    H, W = 64, 96
    img_a = (torch.rand(1, 3, H, W, generator=g) * 2 - 1).to(device)
    img_b = (torch.rand(3, 3, H, W, generator=g) * 2 - 1).to(device)

    def work(img):
        f = model._encode_image(img)[0]
        r1, r2 = model.decode_pair(f, f.flip(0), H, W)
        return f, r1["pts3d"], r2["desc"]

    want_a, want_b = work(img_a), work(img_b)
    torch.cuda.synchronize()
    got = {}

    def runner(name, img, reps):
        with torch.cuda.stream(torch.cuda.Stream(device=device)):
            for _ in range(reps):
                got[name] = work(img)
            torch.cuda.current_stream().synchronize()

    ta = threading.Thread(target=runner, args=("a", img_a, 6))
    tb = threading.Thread(target=runner, args=("b", img_b, 3))
    ta.start(); tb.start(); ta.join(); tb.join()
    for w_, g_ in zip(want_a, got["a"]):
        assert torch.equal(w_, g_)
    for w_, g_ in zip(want_b, got["b"]):
        assert torch.equal(w_, g_)


_FULL = {}


def _full_size_pair(device):
    """The configuration the bench runs - ViT-L encoder (24 x 1024, 16 heads), 12-layer 768-wide decoder, DPT +
    descriptor heads, one 384x512 pair, seeded random weights - through the torch-fp32 oracle on the host and through the
    WHOLE chain on the device (the HIP decoder is fed the HIP encoder's own tokens).  Computed once per session (the
    oracle needs ~10 s of host time)."""
    if not _FULL:
        cfg = R.Mast3rConfig()
        sd, model = _model(cfg, 1, device)
        g = torch.Generator().manual_seed(4)
        H, W = 384, 512
        img1 = torch.rand(1, 3, H, W, generator=g) * 2 - 1
        img2 = torch.rand(1, 3, H, W, generator=g) * 2 - 1
        with torch.inference_mode():
            f1, p1 = R.encode_image(sd, cfg, img1)
            f2, p2 = R.encode_image(sd, cfg, img2)
            d1, d2 = R.decoder(sd, cfg, f1, p1, f2, p2)
            ref1 = R.downstream_head(sd, cfg, 1, d1, H, W)
            ref2 = R.downstream_head(sd, cfg, 2, d2, H, W)
        hf1 = model._encode_image(img1.to(device))[0]
        hf2 = model._encode_image(img2.to(device))[0]
        r1, r2, t1, t2 = model.decode_pair(hf1, hf2, H, W, return_tokens=True)
        _FULL.update(H=H, W=W, f1=f1.numpy(), f2=f2.numpy(), d1=d1[-1].numpy(), d2=d2[-1].numpy(),
                     ref1={k: v.numpy() for k, v in ref1.items()}, ref2={k: v.numpy() for k, v in ref2.items()},
                     hf1=hf1.cpu().numpy(), hf2=hf2.cpu().numpy(), t1=t1.cpu().numpy(), t2=t2.cpu().numpy(),
                     r1={k: v.clone() for k, v in r1.items()}, r2={k: v.clone() for k, v in r2.items()})
        del model
    return _FULL


def test_full_size_model_matches_oracle(device):
    """Full-size forward against the torch-fp32 oracle, fully chained on the device: encoder tokens, then decoder tokens
    and head outputs computed FROM THE HIP ENCODER'S tokens (36 transformer layers of bf16 MFMA operands in a row).
    Tolerances as stated at the top of this file."""
    F = _full_size_pair(device)
    e_enc = max(_rel(F["hf1"], F["f1"]), _rel(F["hf2"], F["f2"]))
    e_dec = max(_rel(F["t1"], F["d1"]), _rel(F["t2"], F["d2"]))
    print(f"full-size rel-L2, chained: encoder tokens {e_enc:.4f}, decoder tokens {e_dec:.4f}")
    assert e_enc <= 2e-2 and e_dec <= 2e-2
    _check_heads(F["r1"], F["ref1"])
    _check_heads(F["r2"], F["ref2"])


def _oracle_match(X11, X21, D11, D21, H, W):
    import oracle
    from oracle import matching_py
    from mast3r_slam.config import config

    mc = config["matching"]
    rays, pts, p0 = matching_py.prep_for_iter_proj(X11, X21)
    p, conv = oracle.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"])
    p1i, v = matching_py.occlusion_and_trunc(X11, X21, p, conv, mc["dist_thresh"])
    p1i = oracle.refine_matches(D11.astype(np.float16), D21.reshape(1, H * W, -1).astype(np.float16), p1i, mc["radius"],
                                mc["dilation_max"])
    return matching_py.pixel_to_lin(p1i, W)[0], v[0]


def _index_report(idx, idx_ref, W, mask=None):
    same = idx == idx_ref
    dist = np.maximum(np.abs(idx % W - idx_ref % W), np.abs(idx // W - idx_ref // W))
    if mask is not None:
        same, dist = same[mask], dist[mask]
    q = (lambda a, t: float(np.quantile(a, t)) if a.size else 0.0)
    moved = dist[~same]
    return float(same.mean()), float((dist <= 1).mean()), q(moved, 0.5), q(moved, 0.9), q(moved, 0.99)


def test_full_size_chain_report_with_random_weights(device):
    """A REPORT, not a parity statement: the whole chain on the device (HIP encode -> HIP decode + heads ->
    matching.match) beside the whole chain of the oracle on one 384x512 pair with seeded RANDOM weights.  Random weights
    give pointmaps without geometry: no pixel passes the reference's 0.1 m occlusion gate on either side (the valid
    flags agree because they are all false) and the projection search is chaotic on such input - only about a fifth
    of the ungated indices coincide and the others are tens of pixels apart.  The asserts are smoke bounds.  What bf16
    does to the integer outputs in a regime where matches EXIST is measured by
    test_match_sensitivity_to_the_network_error_field; on trained weights it cannot be measured offline: parity unpinned."""
    from mast3r_slam import matching

    F = _full_size_pair(device)
    H, W = F["H"], F["W"]
    idx_ref, v_ref = _oracle_match(F["ref1"]["pts3d"], F["ref2"]["pts3d"], F["ref1"]["desc"], F["ref2"]["desc"], H, W)
    idx, valid = matching.match(F["r1"]["pts3d"], F["r2"]["pts3d"], F["r1"]["desc"], F["r2"]["desc"])
    idx, valid = idx[0].cpu().numpy(), valid[0, :, 0].cpu().numpy()
    same, near, med, p90, p99 = _index_report(idx, idx_ref, W)
    print(f"random-weight chain @384x512: valid flags agree {(valid == v_ref).mean():.4f} (valid fraction hip {valid.mean():.4f} / "
          f"oracle {v_ref.mean():.4f}); identical idx {same:.4f}; moved: median {med:.1f} px, p90 {p90:.1f}, p99 {p99:.1f}")
    assert (valid == v_ref).mean() >= 0.98 and same >= 0.10


SENS_SAME_MIN, SENS_NEAR_MIN, SENS_P99_MAX, SENS_VALID_MIN = 0.05, 0.45, 8.0, 0.30     # measured 0.10 / 0.58 / 4 px / 0.41


def test_match_sensitivity_to_the_network_error_field(device):
    """What the bf16 network does to the INTEGER match outputs where matches exist.  The error field of the full-size
    forward - HIP minus oracle, per pixel: relative on the pointmaps (rel-L2 0.015), additive on the unit descriptors
    (cosine 0.9999) - is transplanted onto a pair with real geometry (the procedural room at 384x512: 98 % of the pixels
    pass the occlusion gate): the device's matching.match on the PERTURBED pair against the oracle chain on the clean
    pair.  MEASURED (this is what the bounds below protect, with margin): a 1.4 % relative pointmap error - 4 cm at the
    room's 3 m, against pixels of 7 mm and an occlusion gate of 10 cm - moves nearly every match by about one pixel
    (10 % of the valid pixels keep their index, 58 % stay within one pixel, median move 1 px, p99 4 px) and flips 59 % of
    the valid flags (the 0.1 m gate sits inside the perturbation's radial noise).  So integer match outputs are NOT
    stable under the bf16 error level of a RANDOM-weight network (whose logits feed expm1 at large arguments); what the
    level is on trained weights, and with it match parity, stays unpinned (no checkpoint offline)."""
    from mast3r_slam import matching, synthetic

    F = _full_size_pair(device)
    H, W = F["H"], F["W"]
    pr = synthetic.make_pair(3, 0, h=H, w=W, seed=2)
    clean = {k: pr[k][None].astype(np.float32) for k in ("X11", "X21", "D11", "D21")}
    idx_ref, v_ref = _oracle_match(clean["X11"], clean["X21"], clean["D11"], clean["D21"], H, W)
    pert = {}
    for name, ref, hip in (("X11", F["ref1"]["pts3d"], F["r1"]["pts3d"]), ("X21", F["ref2"]["pts3d"], F["r2"]["pts3d"])):
        rel = (hip.cpu().numpy() - ref) / np.maximum(np.linalg.norm(ref, axis=-1, keepdims=True), 1e-12)   # per pixel, relative
        pert[name] = clean[name] + rel * np.linalg.norm(clean[name], axis=-1, keepdims=True)
    for name, ref, hip in (("D11", F["ref1"]["desc"], F["r1"]["desc"]), ("D21", F["ref2"]["desc"], F["r2"]["desc"])):
        d = clean[name] + (hip.cpu().numpy() - ref)
        pert[name] = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rel_x = _rel(pert["X11"], clean["X11"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
    idx, valid = matching.match(t(pert["X11"]), t(pert["X21"]), t(pert["D11"]), t(pert["D21"]))
    idx, valid = idx[0].cpu().numpy(), valid[0, :, 0].cpu().numpy()
    both = valid & v_ref
    same, near, med, p90, p99 = _index_report(idx, idx_ref, W, both)
    print(f"match sensitivity @384x512: pointmap perturbation rel-L2 {rel_x:.4f}; valid flags agree {(valid == v_ref).mean():.4f} "
          f"(valid fraction {v_ref.mean():.3f}); among valid pixels identical idx {same:.4f}, within 1 px {near:.4f}; "
          f"moved: median {med:.1f} px, p90 {p90:.1f}, p99 {p99:.1f}")
    assert 0.005 <= rel_x <= 0.03
    assert (valid == v_ref).mean() >= SENS_VALID_MIN and same >= SENS_SAME_MIN and near >= SENS_NEAR_MIN and p99 <= SENS_P99_MAX
