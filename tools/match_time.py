#!/usr/bin/env python3
"""The three matching kernels on full-resolution room pairs, each alone on the GPU, event-timed, with checksums of their
outputs (to compare builds).
    python tools/match_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mast3r_slam_backends as be
from mast3r_slam import matching
from mast3r_slam.config import config
from mast3r_slam.synthetic_gpu import RoomRenderer

dev = torch.device("cuda:0")
H, W = 384, 512
R = RoomRenderer(dev, H, W)
mc = config["matching"]
for k in (3.0, 24.0):
    a, b = R.pair_fused(torch.tensor([k], device=dev), torch.tensor([0.0], device=dev))
    rays, pts, p0 = matching.prep_for_iter_proj(a["pts3d"], b["pts3d"], None)
    p_new, conv = be.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"])
    p1 = p_new.long().contiguous()
    D11 = a["desc"].half().contiguous()
    D21 = b["desc"].reshape(1, H * W, -1).half().contiguous()
    def timed(fn, n=20):
        for rep in range(3):
            out = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for rep in range(n):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n, out
    t_prep, _ = timed(lambda: matching.prep_for_iter_proj(a["pts3d"], b["pts3d"], None))
    t_ip, (pn, cv) = timed(lambda: be.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"]))
    t_rf, (out,) = timed(lambda: be.refine_matches(D11, D21, p1, mc["radius"], mc["dilation_max"]))
    print(f"pair ({k:.0f}, 0): prep {t_prep:6.1f} us | iter_proj {t_ip:6.1f} us  sum(p) {float(pn.double().sum()):.6f} converged {int(cv.sum())} | "
          f"refine_matches {t_rf:7.1f} us  checksum {int(out.sum())} moved {float((out != p1).any(-1).float().mean()):.3f}", flush=True)
