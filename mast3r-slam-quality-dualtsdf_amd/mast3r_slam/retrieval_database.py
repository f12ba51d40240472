"""The compute part of mast3r_slam/retrieval_database.py that does not need ASMK: `quantize_custom`
(lines 96-105) - nearest `multiple_assignment` codebook centroids of every local feature by L2 distance, the
(n x d) . (d x 64k) distance GEMM + top-k the reference runs in fp32 torch.

The rest of the class (prep_features: whitening / attention of the retrieval model; ASMK aggregate, inverted file,
Hamming-kernel search) needs the `asmk` package, the retrieval checkpoint and its codebook, none of which is available
offline; `RetrievalDatabase` below therefore takes those parts as injected objects with the reference's call signatures
and only implements what is ours.  SlamSystem accepts any object with `update(frame, add_after_query, k, min_thresh)`.

On the matrix cores in fp32 quality: the bf16 GEMM (csrc/gemm_kernel.h) is run on a hi/lo split of both operands
(q = q_hi + q_lo, three products hi.hi + hi.lo + lo.hi, fp32 accumulate: ~2^-16 relative), which ranks a candidate set of
k + 16 centroids per feature; their distances are then recomputed in fp32 exactly as the reference forms them and the
final top-k is taken from those - index-exact against the reference unless two centroids are closer than fp32 rounding."""
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


class RetrievalDatabase:
    """retrieval_database.py:9-166 with the ASMK-dependent parts injected (`prep_features`, `kernel`, `ivf`, `params`
    as the reference's `Retriever` / `asmk` objects provide them)."""

    def __init__(self, centroids, prep_features=None, asmk_params=None, ivf_builder=None, device="cuda"):
        self.query_device = device
        self.index = CentroidIndex(centroids.to(device))
        self.centroids = self.index.centroids
        self._prep, self.params, self.ivf_builder = prep_features, asmk_params, ivf_builder
        self.kf_counter = 0
        self.kf_ids = []

    def quantize_custom(self, qvecs, params):
        """retrieval_database.py:96-105 -> indices (n, multiple_assignment) int64."""
        return self.index.nearest(qvecs.to(self.query_device), int(params["quantize"]["multiple_assignment"]))

    def update(self, frame, add_after_query, k, min_thresh=0.0):
        raise RuntimeError("RetrievalDatabase.update needs the asmk package, the retrieval checkpoint and its codebook "
                           "(aggregate / inverted file / Hamming search, retrieval_database.py:43-166); none is available "
                           "offline - pass SlamSystem another retriever (e.g. synthetic_gpu.PoseProximityRetriever)")
