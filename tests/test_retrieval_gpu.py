"""RetrievalDatabase.quantize_custom (retrieval_database.py:96-105) on the MFMA GEMM against indices
recorded from the reference method's own source (tests/golden/retrieval_quantize.npz; inputs regenerated from the seed):
65 536 x 1024 codebook, 768 features.  Index-exact.

The whole database (prep_features, quantisation, ASMK aggregation + binarisation, inverted file, Hamming-kernel search,
the update loop) against (a) tests/golden/retrieval_asmk.npz - the reference's own methods + asmk modules + compiled
hamming extension on a small seeded sequence - and (b) the oracle (oracle/asmk_py.py, pinned by the same fixture) at the
published sizes: 65 536 x 1024 codebook, 768 tokens -> 300 local descriptors, 1024-bit signatures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_quantize_custom_matches_reference_indices(device, golden_dir):
    from mast3r_slam.retrieval_database import RetrievalDatabase

    fx = np.load(os.path.join(golden_dir, "retrieval_quantize.npz"))
    g = torch.Generator().manual_seed(int(fx["seed"]))
    centroids = torch.randn(65536, 1024, generator=g)
    q = torch.randn(768, 1024, generator=g)
    q[:64] = centroids[1000:1064] + 0.05 * torch.randn(64, 1024, generator=g)
    db = RetrievalDatabase(None, centroids, device=device)
    for name, k in (("query", 5), ("build", 1)):
        idx = db.quantize_custom(q.to(device), {"quantize": {"multiple_assignment": k}})
        assert idx.shape == (768, k) and idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy(), fx[name])



def _weights(g, device):
    from mast3r_slam.retrieval_database import RetrievalWeights

    return RetrievalWeights((g["pre_m"], g["pre_p"]), [(g["proj_w"], g["proj_b"])], (g["post_m"], g["post_p"]),
                            nfeat=int(g["nfeat"]), device=device)


def test_update_sequence_matches_the_reference(device, golden_dir):
    """Small sizes (descriptor dim 64 = 2 signature words: the scalar popcount path), 12 images, 3 revisits."""
    import types

    from mast3r_slam.retrieval_database import RetrievalDatabase

    g = np.load(os.path.join(golden_dir, "retrieval_asmk.npz"))
    db = RetrievalDatabase(_weights(g, device), torch.from_numpy(g["centroids"]), device=device)
    feats = torch.from_numpy(g["feats"]).to(device)
    n = feats.shape[0]
    for i in range(n):
        frame = types.SimpleNamespace(feat=feats[i])
        # the projector is accumulated in fp64 here (exact products, one rounding); the fixture is the reference's fp32 sgemm
        # over K = 1024, whose own accumulation error is a few ulp: 1e-6 relative
        np.testing.assert_allclose(db.prep_features(frame.feat)[0].cpu().numpy(), g[f"local_{i}"], rtol=1e-6, atol=2e-6)
        inds = db.update(frame, True, 3, 0.005)
        assert inds == g[f"inds_{i}"].tolist(), i
        if i > 0:
            # fp64 sums in the reference's order; the only freedom is the rounding of sim^3 to fp32 (numpy's powf)
            np.testing.assert_allclose(db.last_scores.cpu().numpy(), g[f"scores_{i}"], rtol=2e-7, atol=1e-12)
    probe = types.SimpleNamespace(feat=torch.from_numpy(g["probe_feat"]).to(device))
    assert db.update(probe, False, 4, 0.0) == g["probe_inds"].tolist()
    np.testing.assert_allclose(db.last_scores.cpu().numpy(), g["probe_scores"], rtol=2e-7, atol=1e-12)
    assert db.kf_counter == n and db.n_images == n
    # inverted file, bit for bit (the fixture lists the reference's per-word lists word by word)
    ne = db._starts[-1]
    words = db._e_word[:ne].cpu().numpy().astype(np.int64)
    imids = np.repeat(np.arange(n), np.diff(db._starts))
    sigs = db._e_sig[:ne].cpu().numpy().view(np.uint32)
    order = np.lexsort((imids, words))
    np.testing.assert_array_equal(words[order], g["ivf_words"])
    np.testing.assert_array_equal(imids[order], g["ivf_imids"])
    np.testing.assert_array_equal(sigs[order], g["ivf_vecs"])
    np.testing.assert_array_equal(np.diff(db._starts).astype(np.float64), g["norm_factor"])


def test_published_sizes_against_the_oracle(device):
    """64k x 1024 codebook, ViT-L token shape (768 x 1024), 300 local descriptors, 30 images of which 6 revisit earlier
    ones.  Descriptors and quantisation codes are taken from the device (their own tests pin them; a fifth-nearest centroid
    of a random descriptor is a near-tie in fp32), everything after - aggregation, signatures, inverted file, search,
    selection - is compared with the oracle on identical codes."""
    import types

    from oracle import asmk_py
    from mast3r_slam.retrieval_database import RetrievalDatabase, RetrievalWeights

    gen = torch.Generator().manual_seed(5)
    BD = D = 1024
    K, NT, NIMG = 65536, 768, 30
    eye = torch.eye(BD, dtype=torch.float64)
    w = RetrievalWeights((0.05 * torch.randn(1, BD, generator=gen, dtype=torch.float64),
                          eye + 0.02 * torch.randn(BD, BD, generator=gen, dtype=torch.float64)),
                         [(torch.randn(D, BD, generator=gen) / BD ** 0.5, 0.02 * torch.randn(D, generator=gen))],
                         (0.05 * torch.randn(1, D, generator=gen, dtype=torch.float64),
                          eye + 0.02 * torch.randn(D, D, generator=gen, dtype=torch.float64)), nfeat=300, device=device)
    centroids = torch.randn(K, D, generator=gen)
    db = RetrievalDatabase(w, centroids, device=device)
    ref = asmk_py.RetrievalDatabase(None, centroids.numpy())
    feats = torch.randn(NIMG, 1, NT, BD, generator=gen)
    revisit = {9: 2, 14: 5, 19: 9, 22: 0, 25: 14, 29: 3}
    for new, old in revisit.items():
        feats[new] = feats[old] + 0.08 * torch.randn(1, NT, BD, generator=gen)
    codes = {}
    orig = db.quantize_custom

    def spy(qvecs, params):
        out = orig(qvecs, params)
        codes["last"] = out
        return out

    db.quantize_custom = spy
    quant_agree = []
    for i in range(NIMG):
        frame = types.SimpleNamespace(feat=feats[i].to(device))
        local = db.prep_features(frame.feat)[0]
        assert local.shape == (300, D)
        inds = db.update(frame, True, 3, 0.0)
        c = codes["last"].cpu().numpy()
        want = ref.update_local(local.cpu().numpy(), True, 3, 0.0, codes=c)
        if i in (1, 15):
            quant_agree.append(np.mean(asmk_py.quantize(local.cpu().numpy(), centroids.numpy(), c.shape[1]) == c))
        if i > 0:
            got, exp = db.last_scores.cpu().numpy(), ref.last_scores
            np.testing.assert_allclose(got, exp, rtol=2e-7, atol=1e-13)
            gap = np.sort(exp)[::-1]
            if len(gap) <= 3 or gap[2] - gap[3] > 1e-9:          # the three best are separated: same selection
                assert sorted(inds) == sorted(want), i
            if i in revisit:
                assert inds[0] == revisit[i]                       # the revisited image ranks first
    assert min(quant_agree) >= 0.995, quant_agree
    ne = db._starts[-1]
    np.testing.assert_array_equal(db._e_word[:ne].cpu().numpy(), ref.ivf.words)
    np.testing.assert_array_equal(db._e_sig[:ne].cpu().numpy().view(np.uint32), ref.ivf.vecs)
    np.testing.assert_array_equal(np.diff(db._starts), np.bincount(ref.ivf.imids))


@pytest.mark.parametrize("backend", ["inline", "thread"])
def test_slam_system_with_the_retrieval_database(device, monkeypatch, backend):
    """The product loop with the retrieval class in the retriever slot.  The stand-in encoder writes a smooth code of the
    camera-path position into the tokens (random retrieval head and codebook on top), the camera goes out and comes back:
    on the way back the database must propose keyframes from the way out, and the factor graph must gain loop edges
    between keyframes that are close on the path but far apart in time."""
    from mast3r_slam.config import config
    from mast3r_slam.retrieval_database import RetrievalDatabase, RetrievalWeights
    from mast3r_slam.slam_system import SlamSystem
    from tests.test_slam_system_gpu import RoomModel, _frames

    class PlaceModel(RoomModel):
        def _encode_image(self, img, true_shape=None):
            feat, pos, _ = super()._encode_image(img, true_shape)
            k = feat[:, :1, :1].clone()
            c = torch.arange(1024, device=feat.device, dtype=torch.float32)[None, None, :]
            j = torch.arange(feat.shape[1], device=feat.device, dtype=torch.float32)[None, :, None]
            feat = torch.cos(k * 0.02 * (1.0 + torch.remainder(c, 7.0)) + 1.7 * j + 0.37 * c)
            feat[:, :, 0] = k[:, :, 0]                     # the decoder stand-in reads the path index here
            return feat.contiguous(), pos, None

    monkeypatch.setitem(config["tracking"], "match_frac_thresh", 0.72)
    gen = torch.Generator().manual_seed(3)
    D, K = 128, 2048
    w = RetrievalWeights(None, [(torch.randn(D, 1024, generator=gen) / 32.0, torch.zeros(D))], None, nfeat=24, device=device)
    db = RetrievalDatabase(w, torch.randn(K, D, generator=gen) * 0.7, device=device)
    ks = list(range(0, 63, 3)) + list(range(60, -3, -3))
    torch.manual_seed(0)
    system = SlamSystem(PlaceModel(device), device, retriever=db, frame_group=1, backend=backend)
    frames = _frames(ks, device)
    system.run(frames)
    system.shutdown()
    n_kf = len(system.keyframes)
    assert db.kf_counter == n_kf and n_kf >= 5
    ii, jj = system.factor_graph.ii.cpu().numpy(), system.factor_graph.jj.cpu().numpy()
    kf_k = np.array([ks[system.keyframes[i].frame_id] for i in range(n_kf)])
    loops = [(a, b) for a, b in zip(ii, jj) if abs(int(a) - int(b)) > 2]
    assert loops, (ii, jj)                                  # retrieval added non-consecutive edges ...
    assert any(abs(kf_k[a] - kf_k[b]) <= 12 for a, b in loops), [(kf_k[a], kf_k[b]) for a, b in loops]   # ... of the same place


def test_from_checkpoint_reads_the_reference_file_layout(device, tmp_path, golden_dir):
    """`<name>_trainingfree.pth` = {'args': Namespace(nfeat, residual, ...), 'model': RetrievalModel.state_dict()} next to
    `<name>_codebook.pkl` = Codebook.state_dict() (retrieval/processor.py:64-98, asmk/codebook.py:67-78): files of that
    layout written here from the fixture's tensors give the same database as constructing it from the tensors."""
    import argparse
    import pickle
    import types

    from mast3r_slam.mast3r_utils import load_retriever
    from mast3r_slam.retrieval_database import RetrievalDatabase

    g = np.load(os.path.join(golden_dir, "retrieval_asmk.npz"))
    sd = {"prewhiten.m": torch.from_numpy(g["pre_m"]), "prewhiten.p": torch.from_numpy(g["pre_p"]),
          "projector.0.weight": torch.from_numpy(g["proj_w"]), "projector.0.bias": torch.from_numpy(g["proj_b"]),
          "postwhiten.m": torch.from_numpy(g["post_m"]), "postwhiten.p": torch.from_numpy(g["post_p"])}
    args = argparse.Namespace(nfeat=int(g["nfeat"]), residual=False, nclusters=512, imsize=512)
    pth = tmp_path / "MASt3R_test_retrieval_trainingfree.pth"
    torch.save({"args": args, "model": sd}, pth)
    with open(tmp_path / "MASt3R_test_retrieval_codebook.pkl", "wb") as fh:
        pickle.dump({"type": "Codebook", "params": {"size": 512}, "state": {"centroids": g["centroids"]}}, fh)
    db = load_retriever(None, str(pth), device=device)
    assert isinstance(db, RetrievalDatabase) and db.centroids.shape == (512, 64) and db.weights.nfeat == int(g["nfeat"])
    feats = torch.from_numpy(g["feats"]).to(device)
    for i in range(6):
        assert db.update(types.SimpleNamespace(feat=feats[i]), True, 3, 0.005) == g[f"inds_{i}"].tolist()
