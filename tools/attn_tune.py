#!/usr/bin/env python3
"""Times the attention kernel at the network's shapes under MSLAM_ATTN_SPLIT (one process per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"), os.path.join(ROOT, "tools")]
from bench_kernels import attn
if __name__ == "__main__":
    print("split", os.environ.get("MSLAM_ATTN_SPLIT", "auto"))
    for B, H in ((1, 16), (1, 12), (2, 12), (4, 12), (8, 12), (8, 16)):
        attn(B, H, 768)
