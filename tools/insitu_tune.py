#!/usr/bin/env python3
"""In-situ tile tuning: for every plain GEMM shape of the network at a frame group of 4 (M = 3072) and at the batch-8
backend call (M = 6144), time the WHOLE stage that contains it (encode / pair decode) under each tile configuration
(mslam_gemm_tile_override) and keep the best - isolated warm loops over one shape mispredict.  Prints
`M N K cfg stage_ms_before stage_ms_after` for every shape whose best differs from the built-in choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
H, W = 384, 512
L = m.lib()
CFGS = [642, 643, 644, 1262, 1263, 1242, 1282, 1283, 2128, 2256]
ENC = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 768)]
DEC = [(768, 1024), (2304, 768), (768, 768), (1536, 768), (3072, 768), (768, 3072), (7168, 1792), (6400, 7168)]


def stage(kind, B):
    if kind == "enc":
        img = torch.rand(B, 3, H, W, device=dev) * 2 - 1
        return lambda: model._encode_image(img)
    f = torch.randn(B, 768, mc.enc_dim, device=dev)
    return lambda: model.decode_pair(f, f, H, W)


def timeit(fn, n=6, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


plan = [("enc", 4, ENC), ("dec", 4, DEC), ("dec", 8, DEC)]
B_only = [int(a) for a in sys.argv[1:]] or None
for kind, B, shapes in plan:
    if B_only and B not in B_only:
        continue
    fn = stage(kind, B)
    fn(); fn()
    M = 768 * B
    for N, K in shapes:
        base = timeit(fn)
        best_cfg, best_t = 0, base
        for cfg in CFGS:
            L.mslam_gemm_tile_override(M, N, K, cfg)
            try:
                fn()
                t = timeit(fn)
            except Exception:
                t = 1e9
            if t < best_t * 0.995:
                best_cfg, best_t = cfg, t
        L.mslam_gemm_tile_override(M, N, K, best_cfg)
        tag = "KEEP" if best_cfg == 0 else "BEST"
        print(f"{tag} {kind}{B} {M} {N} {K} cfg {best_cfg} stage {base:.3f} -> {best_t:.3f} ms", flush=True)
    print(f"== {kind}{B} final {timeit(fn):.3f} ms", flush=True)
