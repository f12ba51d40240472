"""ORACLE (test infrastructure only - never imported by the product path): NumPy restatement of the retrieval database
of the reference, SURVEY §8f-1:

  mast3r_slam/retrieval_database.py:24-166        prep_features / update / query / add_to_database / quantize_custom /
                                                  accumulate_scores / add_to_ivf_custom
  thirdparty/mast3r/mast3r/retrieval/model.py     Whitener.forward :62-77, how_select_local :89-105, build_projector :139-151
  thirdparty/mast3r/mast3r/retrieval/processor.py ASMK parameters :84-89
  thirdparty/mast3r/asmk/asmk/kernel.py           aggregate_image :28-42, similarity :59-71
  thirdparty/mast3r/asmk/asmk/functional.py       asmk_kernel :10-15
  thirdparty/mast3r/asmk/asmk/inverted_file.py    IVF.add :61-88, IVF.search :90-114
  thirdparty/mast3r/asmk/cython/hamming.pyx       binarize_and_pack_2D :93-127, hamming_cdist_packed :135-152

Pinned by tests/golden/retrieval_asmk.npz (the reference's own methods + ASMKKernel + IVF + compiled hamming extension run
on seeded inputs, tests/golden/make_golden.py section retrieval_asmk) and tests/golden/asmk_hamming.npz (the extension's
outputs on the inputs of its own unit tests' kind).  The inverted file is kept as flat arrays (entries in insertion order,
i.e. image by image, words ascending inside an image) instead of the reference's per-word python lists; `search` walks
the query words in ascending order like the reference, so the float64 accumulation order per image is the same."""
import numpy as np

ASMK_PARAMS = {     # processor.py:84-89 (configuration values)
    "build_ivf": {"kernel": {"binary": True}, "ivf": {"use_idf": False}, "quantize": {"multiple_assignment": 1},
                  "aggregate": {}},
    "query_ivf": {"quantize": {"multiple_assignment": 5}, "aggregate": {}, "search": {"topk": None},
                  "similarity": {"similarity_threshold": 0.0, "alpha": 3.0}},
}


def binarize_and_pack_2D(arr, threshold=0):
    """hamming.pyx:93-127: bit = arr > threshold, packed MSB-first into uint32, the last word left-aligned."""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    n, d = arr.shape
    nw = int(np.ceil(d / 32.0))
    bits = np.zeros((n, nw * 32), dtype=np.uint64)
    bits[:, :d] = arr > threshold
    w = (bits.reshape(n, nw, 32) << np.arange(31, -1, -1, dtype=np.uint64)).sum(-1)
    return w.astype(np.uint32)


def hamming_cdist_packed(a, b, normalization=0):
    """hamming.pyx:135-152 (c_hamming_dist_uint32_arr :33-41): popcount(a ^ b) / normalization as C float."""
    a, b = np.asarray(a, dtype=np.uint32), np.asarray(b, dtype=np.uint32)
    norm = np.float32(normalization) if normalization != 0 else np.float32(a.shape[1] * 32)
    x = a[:, None, :] ^ b[None, :, :]
    cnt = np.unpackbits(x.view(np.uint8), axis=-1).sum(-1).astype(np.int32)
    return (cnt.astype(np.float32) / norm).astype(np.float32)


def whiten(x, m, p):
    """Whitener.forward with l2norm=None (model.py:62-77): float64 centre + matmul, result cast back to x.dtype."""
    out = (x.reshape(-1, x.shape[-1]).astype(np.float64) - m) @ p
    return out.reshape(x.shape[:-1] + (p.shape[1],)).astype(x.dtype)


def prep_features(backbone_feat, w):
    """retrieval_database.py:24-41.  `w`: dict pre_m, pre_p, proj_w, proj_b, post_m, post_p, nfeat (single Linear projector,
    residual False: the published retrieval checkpoint's layout, hdims=[1024])."""
    x = whiten(backbone_feat, w["pre_m"], w["pre_p"])
    proj = (x @ w["proj_w"].T + w["proj_b"]).astype(np.float32)
    attention = np.linalg.norm(proj, axis=-1)
    post = whiten(proj, w["post_m"], w["post_p"])
    nfeat = int(w["nfeat"])
    if nfeat < 0:
        nfeat = int(-nfeat * post.shape[1])
    k = min(nfeat, attention.shape[1])
    order = np.argsort(-attention, axis=1, kind="stable")[:, :k]          # torch.topk: descending
    return np.take_along_axis(post, order[..., None], axis=1)


def quantize(qvecs, centroids, k):
    """retrieval_database.py:96-105."""
    q, c = qvecs.astype(np.float32), centroids.astype(np.float32)
    d = (q ** 2).sum(1)[:, None] + (c ** 2).sum(1)[None, :] - 2 * (q @ c.T)
    return np.argsort(d, axis=1, kind="stable")[:, :k]


def aggregate_image(des, word_ids, centroids):
    """kernel.py:28-42 (binary=True): per visual word the sum of the residuals of the descriptors assigned to it."""
    unique_ids = np.unique(word_ids)
    ades = np.empty((unique_ids.shape[0], des.shape[1]), dtype=np.float32)
    for i, word in enumerate(unique_ids):
        ades[i] = (des[(word_ids == word).any(axis=1)] - centroids[word]).sum(0)
    return binarize_and_pack_2D(ades), unique_ids


class IVF:
    """inverted_file.py:8-114 with use_idf=False (processor.py:85), flat storage."""

    def __init__(self, n_words_sig):
        self.words = np.zeros(0, dtype=np.int64)
        self.imids = np.zeros(0, dtype=np.int64)
        self.vecs = np.zeros((0, n_words_sig), dtype=np.uint32)
        self.norm_factor = np.zeros(0, dtype=np.float64)
        self.n_images = 0

    def add(self, des, word_ids, image_ids):
        """:61-88."""
        assert image_ids.min() >= self.n_images
        top = int(image_ids.max()) + 1
        self.norm_factor = np.concatenate((self.norm_factor, np.zeros(top - len(self.norm_factor))))
        self.n_images = max(self.n_images, top)
        self.words = np.concatenate((self.words, word_ids.astype(np.int64)))
        self.imids = np.concatenate((self.imids, image_ids.astype(np.int64)))
        self.vecs = np.concatenate((self.vecs, des))
        np.add.at(self.norm_factor, image_ids, 1.0)

    def search(self, des, word_ids, alpha, similarity_threshold):
        """:90-114 + kernel.py:59-71 + functional.py:10-15; returns the scores in image order (the reference returns them
        ranked and retrieval_database.py:60-62 undoes the ranking)."""
        scores = np.zeros(self.n_images)
        q_norm_factor = 0
        for qvec, word in zip(des, word_ids):
            q_norm_factor += np.float32(1.0)                       # idf[word], all ones without idf
            sel = np.nonzero(self.words == word)[0]
            if len(sel) == 0:
                continue
            norm_hdist = hamming_cdist_packed(qvec.reshape(1, -1), self.vecs[sel])
            sim = -2 * norm_hdist.squeeze(0) + 1
            mask = sim >= similarity_threshold
            image_ids = self.imids[sel][mask]
            sim = np.power(sim[mask], alpha)
            sim *= np.float32(1.0)
            sim /= np.sqrt(self.norm_factor[image_ids])
            scores[image_ids] += sim
        return scores / np.sqrt(q_norm_factor)


class RetrievalDatabase:
    """retrieval_database.py:9-166 on the pieces above."""

    def __init__(self, weights, centroids, params=None):
        self.w, self.centroids = weights, np.asarray(centroids, dtype=np.float32)
        self.params = params or ASMK_PARAMS
        self.ivf = IVF(int(np.ceil(self.centroids.shape[1] / 32.0)))
        self.kf_counter, self.kf_ids = 0, []
        self.last_scores = None

    def update(self, feat, add_after_query, k, min_thresh=0.0):
        """:43-75; `feat` = frame.feat (1, tokens, backbone dim)."""
        return self.update_local(prep_features(feat, self.w)[0], add_after_query, k, min_thresh)

    def update_local(self, local, add_after_query, k, min_thresh=0.0, codes=None):
        """The same from the local descriptors on (`codes`: optional precomputed quantize() output with the query's
        multiple assignment, for tests that pin the later stages on identical assignments)."""
        inds, topk_codes = [], None
        if self.kf_counter > 0:
            q = self.params["query_ivf"]
            topk_codes = quantize(local, self.centroids, q["quantize"]["multiple_assignment"]) if codes is None else codes
            ades, uniq = aggregate_image(local, topk_codes, self.centroids)
            scores = self.ivf.search(ades, uniq, **q["similarity"])
            self.last_scores = scores
            kk = min(k, self.ivf.n_images)
            order = np.argsort(-scores, kind="stable")[:kk]
            inds = [int(i) for i in order if scores[i] > min_thresh]
        if add_after_query:
            kb = self.params["build_ivf"]["quantize"]["multiple_assignment"]
            if topk_codes is None:
                topk_codes = quantize(local, self.centroids, kb) if codes is None else codes
            ades, uniq = aggregate_image(local, topk_codes[:, :kb], self.centroids)
            self.ivf.add(ades, uniq, np.full(len(uniq), self.kf_counter, dtype=np.int64))
            self.kf_ids.append(self.kf_counter)
            self.kf_counter += 1
        return inds
