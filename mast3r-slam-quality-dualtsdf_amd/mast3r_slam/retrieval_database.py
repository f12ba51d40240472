"""Mirror of mast3r_slam/retrieval_database.py (lines 9-166): the keyframe retrieval database - MASt3R retrieval head
(whitening, projector, attention top-k), ASMK with binarised residuals over a 64k codebook, incremental inverted file -
with the same methods (`prep_features`, `update`, `query`, `add_to_database`, `quantize_custom`) and the same results,
resident on the device.

What runs where
  * `quantize_custom` (lines 96-105): the (n x d) . (d x 64k) distance GEMM + top-k on the bf16 MFMA GEMM
    (csrc/gemm_kernel.h) in fp32 quality - hi/lo split of both operands (three products, ~2^-16 relative) ranks a
    candidate set of k + 16 centroids per feature, whose distances are then recomputed in fp32 exactly as the reference
    forms them; index-exact against the reference unless two centroids are closer than fp32 rounding.
  * aggregation + binarisation (asmk kernel.py:28-42, hamming.pyx:93-127) and the inverted-file search with the Hamming
    kernel (inverted_file.py:90-114, kernel.py:59-71, functional.py:10-15): csrc/retrieval.hip.  The reference keeps a
    python list of numpy arrays per visual word and walks the query's words in a python loop; here the file is three
    flat device arrays in insertion order (word id, 1024-bit signature, image offsets) and one launch scores every
    database image (a block per image, its entries summed in the reference's order, so scores agree to the last bit of
    the fp64 accumulation apart from the rounding of sim^3).
  * `prep_features` (lines 24-41): the two fp64 whitening GEMMs and the projector's Linear layers (768 x 1024 x 1024) on the
    f64 matrix cores (csrc/retrieval.hip::gemm_f64_kernel), attention = row norm, top-`nfeat` rows.

The retrieval checkpoint and its codebook pickle are not available offline: `RetrievalDatabase.from_checkpoint` follows
thirdparty/mast3r/mast3r/retrieval/processor.py:64-98 for whoever has the files; tests and the synthetic runs construct the
class from tensors (`RetrievalWeights`, a codebook).  SlamSystem accepts any object with
`update(frame, add_after_query, k, min_thresh)`."""
import os
import pickle

import torch

import mslam_hip as _m


def _split_bf16(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def _gemm_nt(a_bf16, w_bf16):
    """out f32[M,N] = a[M,K] . w[N,K]^T on the bf16 MFMA GEMM."""
    M, K = a_bf16.shape
    N = w_bf16.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a_bf16.device)
    rc = _m.lib().mslam_gemm_bf16(_m.ptr(a_bf16), _m.ptr(w_bf16), 0, 0, _m.ptr(out), M, N, K, 0, 0, _m.stream_ptr())
    _m.check(rc, "gemm_bf16")
    return out


class CentroidIndex:
    """The codebook prepared once: hi/lo bf16 halves and squared norms."""

    def __init__(self, centroids):
        self.centroids = centroids.float().contiguous()
        self.hi, self.lo = _split_bf16(self.centroids)
        self.sq = torch.sum(self.centroids ** 2, dim=1)

    @torch.inference_mode()
    def nearest(self, qvecs, k, margin=16):
        q = qvecs.float().contiguous()
        q_hi, q_lo = _split_bf16(q)
        dot = _gemm_nt(q_hi, self.hi)
        dot += _gemm_nt(q_hi, self.lo)
        dot += _gemm_nt(q_lo, self.hi)
        qsq = torch.sum(q ** 2, dim=1)
        coarse = qsq[:, None] + self.sq[None, :] - 2 * dot
        kk = min(k + margin, self.centroids.shape[0])
        cand = torch.topk(coarse, kk, dim=1, largest=False).indices                     # (n, kk)
        c = self.centroids[cand]                                                         # (n, kk, d) fp32
        exact = qsq[:, None] + self.sq[cand] - 2 * torch.einsum("nd,nkd->nk", q, c)      # the reference's formula, fp32
        order = torch.topk(exact, k, dim=1, largest=False).indices
        return torch.gather(cand, 1, order)


ASMK_PARAMS = {     # thirdparty/mast3r/mast3r/retrieval/processor.py:84-89
    "build_ivf": {"kernel": {"binary": True}, "ivf": {"use_idf": False}, "quantize": {"multiple_assignment": 1},
                  "aggregate": {}},
    "query_ivf": {"quantize": {"multiple_assignment": 5}, "aggregate": {}, "search": {"topk": None},
                  "similarity": {"similarity_threshold": 0.0, "alpha": 3.0}},
}


class RetrievalWeights:
    """The retrieval head of thirdparty/mast3r/mast3r/retrieval/model.py:108-151 without its backbone: `prewhiten` /
    `postwhiten` = (m (1, d) f64, p (d, d') f64) or None (nn.Identity), `projector` = list of Linear layers
    (weight, bias[, ln_weight, ln_bias]) - every layer but the last is followed by LayerNorm + GELU (build_projector) -
    `residual`, `nfeat`."""

    def __init__(self, prewhiten, projector, postwhiten, nfeat=300, residual=False, device="cuda"):
        f64 = lambda t: None if t is None else tuple(torch.as_tensor(x).to(device=device, dtype=torch.float64) for x in t)
        self.prewhiten, self.postwhiten = f64(prewhiten), f64(postwhiten)
        self.projector = [tuple(torch.as_tensor(x).to(device=device, dtype=torch.float32) for x in layer) for layer in projector]
        self.nfeat, self.residual = nfeat, bool(residual)

    @classmethod
    def from_state_dict(cls, sd, nfeat, residual, device="cuda"):
        """Keys of RetrievalModel.state_dict(): prewhiten.m/p, projector.<i>.weight/bias, postwhiten.m/p."""
        wh = lambda k: (sd[f"{k}.m"], sd[f"{k}.p"]) if f"{k}.m" in sd else None
        ids = sorted({int(k.split(".")[1]) for k in sd if k.startswith("projector.")})
        lin = [i for i in ids if sd[f"projector.{i}.weight"].ndim == 2]
        layers = []
        for i in lin:
            layer = [sd[f"projector.{i}.weight"], sd[f"projector.{i}.bias"]]
            if f"projector.{i + 1}.weight" in sd and sd[f"projector.{i + 1}.weight"].ndim == 1:
                layer += [sd[f"projector.{i + 1}.weight"], sd[f"projector.{i + 1}.bias"]]
            layers.append(tuple(layer))
        return cls(wh("prewhiten"), layers, wh("postwhiten"), nfeat=nfeat, residual=residual, device=device)


def _gemm_f64(a, b, b_transposed, centre=None, bias=None):
    """out f64[M,N] = (a[M,K] - centre[K]) . b (+ bias[N]) on the f64 matrix cores (csrc/retrieval.hip); a, b f32 or f64."""
    a, b = a.contiguous(), b.contiguous()
    M, K = a.shape
    N = b.shape[0] if b_transposed else b.shape[1]
    out = torch.empty((M, N), dtype=torch.float64, device=a.device)
    cen = None if centre is None else centre.reshape(-1).to(torch.float64).contiguous()
    bs = None if bias is None else bias.reshape(-1).to(torch.float64).contiguous()
    rc = _m.lib().mslam_gemm_f64(_m.ptr(a), int(a.dtype == torch.float32), _m.ptr(b), int(b.dtype == torch.float32),
                                 int(bool(b_transposed)), _m.ptr(cen), _m.ptr(bs), _m.ptr(out), M, N, K, _m.stream_ptr())
    _m.check(rc, "gemm_f64")
    return out


def _whiten(x, mp):
    """Whitener.forward (retrieval/model.py:62-77, l2norm None): fp64 centre + matmul, cast back to the input type."""
    if mp is None:
        return x
    m, p = mp
    out = _gemm_f64(x.reshape(-1, x.shape[-1]), p, False, centre=m)
    return out.reshape(x.shape[:-1] + (p.shape[1],)).to(x.dtype)


def _linear(x, weight, bias):
    """nn.Linear of the projector (retrieval/model.py:108-151; fp32 in the reference): exact products of the fp32
    operands accumulated in fp64 on the matrix cores, rounded to fp32 once."""
    out = _gemm_f64(x.reshape(-1, x.shape[-1]), weight, True, bias=bias)
    return out.reshape(x.shape[:-1] + (weight.shape[0],)).to(x.dtype)


class _ArrayOnlyUnpickler(pickle.Unpickler):
    """Unpickler for Codebook.state_dict() files (asmk/codebook.py:67-78: nested dicts of python scalars and one
    float32 ndarray): only the functions numpy's own array pickles name are resolvable, anything else - i.e. every
    `__reduce__` payload - raises before it is called."""

    def find_class(self, module, name):
        import numpy as np

        ma = getattr(np, "_core", None) or np.core          # numpy 2.x / 1.x
        table = {("numpy.core.multiarray", "_reconstruct"): ma.multiarray._reconstruct,
                 ("numpy._core.multiarray", "_reconstruct"): ma.multiarray._reconstruct,
                 ("numpy.core.multiarray", "scalar"): ma.multiarray.scalar,
                 ("numpy._core.multiarray", "scalar"): ma.multiarray.scalar,
                 ("numpy", "ndarray"): np.ndarray, ("numpy", "dtype"): np.dtype}
        if (module, name) in table:
            return table[(module, name)]
        raise pickle.UnpicklingError(f"codebook file names {module}.{name}: only numpy arrays are loaded")


def load_codebook(path):
    """The (K, d) float32 centroid matrix of an ASMK codebook file without executing anything from it: `.pkl` =
    Codebook.state_dict() ({'state': {'centroids': ndarray, ...}, ...}) through _ArrayOnlyUnpickler; `.npz` (key
    'centroids') and `.npy` through numpy.load with allow_pickle=False."""
    import numpy as np

    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path, allow_pickle=False), dtype=np.float32)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return np.ascontiguousarray(z["centroids"], dtype=np.float32)
    with open(path, "rb") as fh:
        obj = _ArrayOnlyUnpickler(fh).load()
    cen = obj["state"]["centroids"] if "state" in obj else obj["centroids"]
    return np.ascontiguousarray(cen, dtype=np.float32)


class RetrievalDatabase:
    """retrieval_database.py:9-166.  `weights`: RetrievalWeights; `centroids`: the ASMK codebook (K, d) f32."""

    def __init__(self, weights, centroids, asmk_params=None, device="cuda"):
        self.query_device = device
        self.query_dtype = torch.float32
        self.weights = weights
        self.index = CentroidIndex(torch.as_tensor(centroids).to(device))
        self.centroids = self.index.centroids
        self.params = asmk_params or ASMK_PARAMS
        if self.params["build_ivf"]["ivf"]["use_idf"] or not self.params["build_ivf"]["kernel"]["binary"]:
            raise ValueError("only the reference's configuration is built: binary signatures, no idf (processor.py:85)")
        dim = int(self.centroids.shape[1])
        if dim % 32:
            raise ValueError(f"descriptor dimension {dim} is not a multiple of 32")
        self.sig_words = dim // 32
        # the inverted file: flat arrays in insertion order, grown by doubling
        self._cap = 0
        self._e_word = torch.zeros(0, dtype=torch.int32, device=device)
        self._e_sig = torch.zeros((0, self.sig_words), dtype=torch.int32, device=device)      # uint32 bit patterns
        self._starts = [0]                               # host mirror of img_start
        self._img_start = torch.zeros(1, dtype=torch.int32, device=device)
        self.kf_counter = 0
        self.kf_ids = []
        self.last_scores = None

    @classmethod
    def from_checkpoint(cls, modelname, device="cuda"):
        """thirdparty/mast3r/mast3r/retrieval/processor.py:64-98: `<name>.pth` = {'args': Namespace, 'model': state_dict},
        `<name minus its last _field>_codebook.pkl` = Codebook.state_dict() (asmk/codebook.py:67-78).  Files supplied by the
        user (they are not part of the reference tree).  The reference unpickles both (torch.load weights_only=False,
        pickle.load); here nothing from either file is executed: the checkpoint goes through torch's weights-only
        loader with argparse.Namespace (plain attribute container) allow-listed, the codebook through an unpickler that
        admits the numpy array reconstructors and nothing else (`load_codebook`; `.npy` / `.npz` are accepted too)."""
        import argparse

        assert os.path.isfile(modelname), modelname
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(modelname, "cpu", weights_only=True)
        a = ckpt["args"]
        dname, bname = os.path.split(modelname)
        stem = os.path.join(dname, "_".join(bname.split("_")[:-1]) + "_codebook")
        cb = next((stem + ext for ext in (".pkl", ".npz", ".npy") if os.path.isfile(stem + ext)), None)
        assert cb is not None, stem + ".pkl"
        centroids = load_codebook(cb)
        nfeat = a.nfeat if hasattr(a, "nfeat") else a["nfeat"]
        residual = getattr(a, "residual", False) if not isinstance(a, dict) else a.get("residual", False)
        w = RetrievalWeights.from_state_dict(ckpt["model"], nfeat=nfeat, residual=residual, device=device)
        return cls(w, torch.from_numpy(centroids), device=device)

    @property
    def n_images(self):
        return len(self._starts) - 1

    # -------------------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def prep_features(self, backbone_feat):
        """:24-41 (retrieval/model.py:205-216 without the encoder; how_select_local :89-105)."""
        w = self.weights
        x = _whiten(backbone_feat.to(self.query_device), w.prewhiten)
        proj = x
        for li, layer in enumerate(w.projector):
            proj = _linear(proj, layer[0], layer[1])
            if li + 1 < len(w.projector):
                proj = torch.nn.functional.gelu(torch.nn.functional.layer_norm(proj, proj.shape[-1:], layer[2], layer[3]))
        if w.residual:
            proj = proj + x
        attention = proj.norm(dim=-1)
        post = _whiten(proj, w.postwhiten)
        nfeat = w.nfeat
        nfeat = int(-nfeat * post.size(1)) if nfeat < 0 else int(nfeat)
        idx = torch.topk(attention, min(nfeat, attention.size(1)), dim=1).indices
        return torch.gather(post, 1, idx.unsqueeze(-1).expand(-1, -1, post.size(2)))

    def quantize_custom(self, qvecs, params):
        """:96-105 -> indices (n, multiple_assignment) int64."""
        return self.index.nearest(qvecs.to(self.query_device), int(params["quantize"]["multiple_assignment"]))

    def _aggregate(self, des, codes):
        """kernel.py:28-42: (signatures u32 bit patterns (U, W) as int32, sorted unique words int64 (U,))."""
        codes = codes.contiguous()
        uniq = torch.unique(codes)                                   # sorted; its length is read by the host here
        sig = torch.empty((uniq.numel(), self.sig_words), dtype=torch.int32, device=des.device)
        rc = _m.lib().mslam_asmk_aggregate(_m.ptr(des), _m.ptr(self.centroids), _m.ptr(codes), _m.ptr(uniq), _m.ptr(sig),
                                           des.shape[0], codes.shape[1], des.shape[1], uniq.numel(),
                                           self.centroids.shape[0], _m.stream_ptr())
        _m.check(rc, "asmk_aggregate")
        return sig, uniq

    @torch.inference_mode()
    def query(self, feat, id=None):
        """:77-93 + accumulate_scores :107-137.  `feat`: local descriptors (n, d) of ONE image.  Returns (scores f64
        (n_images,) in image order, topk codes (n, multiple_assignment)) - the reference's (ranks, ranked scores) pair is
        this vector sorted, and `update` (:58-62) undoes the sorting first thing."""
        q = self.params["query_ivf"]
        des = feat.to(self.query_device, torch.float32).contiguous()
        topk = self.quantize_custom(des, q)
        sig, uniq = self._aggregate(des, topk)
        scores = torch.zeros(self.n_images, dtype=torch.float64, device=des.device)
        uniq32 = uniq.to(torch.int32)        # a named local: the pointer must outlive the launch's enqueue
        rc = _m.lib().mslam_asmk_search(_m.ptr(self._e_word), _m.ptr(self._e_sig), _m.ptr(self._img_start), self.n_images,
                                        _m.ptr(uniq32), _m.ptr(sig), uniq.numel(), self.sig_words,
                                        float(q["similarity"]["similarity_threshold"]), float(q["similarity"]["alpha"]),
                                        _m.ptr(scores), _m.stream_ptr())
        _m.check(rc, "asmk_search")
        return scores, topk

    @torch.inference_mode()
    def add_to_database(self, feat, id=None, topk_codes=None):
        """:95-101 + add_to_ivf_custom :139-166 + IVF.add (inverted_file.py:61-88)."""
        des = feat.to(self.query_device, torch.float32).contiguous()
        b = self.params["build_ivf"]
        kb = int(b["quantize"]["multiple_assignment"])
        codes = self.quantize_custom(des, b) if topk_codes is None else topk_codes[:, :kb]
        sig, uniq = self._aggregate(des, codes)
        n0, n1 = self._starts[-1], self._starts[-1] + uniq.numel()
        if n1 > self._cap:
            cap = max(4096, 2 * self._cap, n1)
            ew = torch.zeros(cap, dtype=torch.int32, device=des.device)
            es = torch.zeros((cap, self.sig_words), dtype=torch.int32, device=des.device)
            ew[:n0], es[:n0] = self._e_word[:n0], self._e_sig[:n0]
            self._e_word, self._e_sig, self._cap = ew, es, cap
        self._e_word[n0:n1] = uniq.to(torch.int32)
        self._e_sig[n0:n1] = sig
        self._starts.append(n1)
        self._img_start = torch.tensor(self._starts, dtype=torch.int32, device=des.device)
        self.kf_ids.append(self.kf_counter)
        self.kf_counter += 1

    @torch.inference_mode()
    def update(self, frame, add_after_query, k, min_thresh=0.0):
        """:43-75."""
        feat = self.prep_features(frame.feat)[0]                     # one frame at a time, as the reference assumes
        topk_image_inds, topk_codes = [], None
        if self.kf_counter > 0:
            scores, topk_codes = self.query(feat)
            self.last_scores = scores
            top = torch.topk(scores, min(k, self.n_images))
            topk_image_inds = top.indices[top.values > min_thresh].tolist()
        if add_after_query:
            self.add_to_database(feat, None, topk_codes)
        return topk_image_inds
