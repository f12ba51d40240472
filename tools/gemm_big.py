#!/usr/bin/env python3
"""Large-shape subset of tools/gemm_tune.py (backend batch, DPT convolutions as dense GEMMs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"), os.path.join(ROOT, "tools")]
from gemm_sweep import gemm
SHAPES = [(3072, 2304, 768), (3072, 3072, 768), (3072, 7168, 1792), (3072, 6400, 7168), (49152, 256, 2304), (49152, 128, 2304),
          (196608, 128, 1152), (4096, 4096, 4096), (8192, 8192, 8192)]
if __name__ == "__main__":
    cfg = os.environ.get("MSLAM_GEMM", "auto")
    for M, N, K in SHAPES:
        us = gemm(M, N, K, iters=10)
        print(f"{cfg} {M} {N} {K} {us:.1f}")
