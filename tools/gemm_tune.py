#!/usr/bin/env python3
"""Times every GEMM shape of the network under the tile/ring configuration given by MSLAM_GEMM
(one process per configuration, see tools/gemm_tune.sh).  Prints 'M N K us' lines."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"), os.path.join(ROOT, "tools")]
from gemm_sweep import gemm

SHAPES = [  # (M, N, K): encoder / decoder / head linears at B=1 and at the 4-edge backend batch
    (768, 3072, 1024), (768, 1024, 1024), (768, 4096, 1024), (768, 1024, 4096),
    (768, 2304, 768), (768, 768, 768), (768, 1536, 768), (768, 3072, 768), (768, 768, 3072), (768, 768, 1024),
    (768, 7168, 1792), (768, 6400, 7168),
    (3072, 2304, 768), (3072, 768, 768), (3072, 1536, 768), (3072, 3072, 768), (3072, 768, 3072), (3072, 768, 1024),
    (3072, 7168, 1792), (3072, 6400, 7168), (8192, 8192, 8192),
]
if os.environ.get("MSLAM_TUNE_SET") == "groups":   # frame groups of 2 / 4 and the batch-8 backend call
    SHAPES = [(M, N, K) for M in (1536, 3072, 6144) for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))]
    SHAPES += [(M, N, K) for M in (1536, 6144)
               for N, K in ((2304, 768), (768, 768), (1536, 768), (3072, 768), (768, 3072), (768, 1024))]
if __name__ == "__main__":
    cfg = os.environ.get("MSLAM_GEMM", "auto")
    for M, N, K in SHAPES:
        try:
            us = gemm(M, N, K, iters=30 if M * N * K < 1e11 else 5)
        except Exception as e:  # configuration not built
            print(f"{cfg} {M} {N} {K} nan"); continue
        print(f"{cfg} {M} {N} {K} {us:.1f}")
