#!/usr/bin/env python3
"""Times matching.match (prep + iter_proj + occlusion + refine_matches + pixel_to_lin) at 384x512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import numpy as np, torch
from mast3r_slam import synthetic, matching
dev = torch.device("cuda:0")
pr = synthetic.make_pair(3, 0, h=384, w=512, seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B in (1, 8):
    X11, X21, D11, D21 = (t(pr[k])[None].expand(B, -1, -1, -1).contiguous() for k in ("X11", "X21", "D11", "D21"))
    for _ in range(3):
        matching.match(X11, X21, D11, D21)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        matching.match(X11, X21, D11, D21)
    e1.record(); torch.cuda.synchronize()
    print(f"match B={B}: {e0.elapsed_time(e1) / 10:.3f} ms")
