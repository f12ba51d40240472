"""Oracle of the retrieval database (oracle/asmk_py.py) against the fixtures the reference's own code produced
(tests/golden/retrieval_asmk.npz, asmk_hamming.npz: make_golden.py section retrieval_asmk) and the known answers its
hamming docstrings state (thirdparty/mast3r/asmk/cython/hamming.pyx:60-61, 117-118, 135-139)."""
import os

import numpy as np
import pytest

from oracle import asmk_py

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "retrieval_asmk.npz"))


def weights(g):
    return {k: g[k] for k in ("pre_m", "pre_p", "proj_w", "proj_b", "post_m", "post_p", "nfeat")}


def test_hamming_docstring_answers():
    u = lambda *rows: np.array(rows, dtype=np.uint32)
    np.testing.assert_array_equal(asmk_py.hamming_cdist_packed(u([3]), u([1]), 2), [[0.5]])
    np.testing.assert_array_equal(asmk_py.hamming_cdist_packed(u([3], [1]), u([1], [2]), 2), [[0.5, 0.5], [0.0, 1.0]])
    # packing == numpy.packbits big-endian words (the property asmk/test/test_hamming.py checks)
    r = np.random.default_rng(0)
    for d in range(1, 70):
        a = (r.random((5, d)) - 0.5).astype(np.float32)
        pb = np.packbits(a > 0, axis=1)
        pb = np.pad(pb, ((0, 0), (0, (-pb.shape[1]) % 4)))
        want = pb.reshape(5, -1, 4).astype(np.uint32) @ np.array([1 << 24, 1 << 16, 1 << 8, 1], dtype=np.uint32)
        np.testing.assert_array_equal(asmk_py.binarize_and_pack_2D(a), want)


def test_hamming_against_the_reference_extension():
    k = np.load(os.path.join(GOLD, "asmk_hamming.npz"))
    for d in (1, 7, 31, 32, 33, 64, 100, 128, 139):
        pa = asmk_py.binarize_and_pack_2D(k[f"a_{d}"])
        np.testing.assert_array_equal(pa, k[f"pack_a_{d}"])
        got = asmk_py.hamming_cdist_packed(pa, asmk_py.binarize_and_pack_2D(k[f"b_{d}"]), d)
        np.testing.assert_array_equal(got, k[f"cdist_{d}"])


def test_prep_features(gold):
    w = weights(gold)
    for i in range(gold["feats"].shape[0]):
        got = asmk_py.prep_features(gold["feats"][i], w)[0]
        np.testing.assert_allclose(got, gold[f"local_{i}"], rtol=0, atol=2e-6)      # fp32 GEMM order (torch vs numpy)


def test_update_sequence(gold):
    db = asmk_py.RetrievalDatabase(weights(gold), gold["centroids"])
    n = gold["feats"].shape[0]
    for i in range(n):
        inds = db.update(gold["feats"][i], True, 3, 0.005)
        assert inds == gold[f"inds_{i}"].tolist(), i
        if i > 0:
            np.testing.assert_allclose(db.last_scores, gold[f"scores_{i}"], rtol=1e-12, atol=1e-15)
    inds = db.update(gold["probe_feat"], False, 4, 0.0)
    assert inds == gold["probe_inds"].tolist()
    np.testing.assert_allclose(db.last_scores, gold["probe_scores"], rtol=1e-12, atol=1e-15)
    assert db.kf_counter == n                                          # the probe was not added
    # inverted file contents: the reference keeps per-word lists, the fixture lists them word by word
    order = np.lexsort((db.ivf.imids, db.ivf.words))
    np.testing.assert_array_equal(db.ivf.words[order], gold["ivf_words"])
    np.testing.assert_array_equal(db.ivf.imids[order], gold["ivf_imids"])
    np.testing.assert_array_equal(db.ivf.vecs[order], gold["ivf_vecs"])
    np.testing.assert_array_equal(db.ivf.norm_factor, gold["norm_factor"])


@pytest.mark.skipif(not os.path.isdir(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "asmk_ext")),
                    reason="oracle/_ref/asmk_ext (the reference's hamming extension, make -C oracle ref) not built")
def test_oracle_against_the_built_reference_extension_directly():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "asmk_ext"))
    try:
        import hamming
    finally:
        sys.path.pop(0)
    r = np.random.default_rng(9)
    a = (r.random((40, 1024)) - 0.5).astype(np.float32)
    b = (r.random((60, 1024)) - 0.5).astype(np.float32)
    pa, pb = hamming.binarize_and_pack_2D(a), hamming.binarize_and_pack_2D(b)
    np.testing.assert_array_equal(asmk_py.binarize_and_pack_2D(a), pa)
    np.testing.assert_array_equal(asmk_py.hamming_cdist_packed(pa, pb), hamming.hamming_cdist_packed(pa, pb))
