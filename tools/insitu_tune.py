#!/usr/bin/env python3
"""In-situ tile tuning: for every GEMM / implicit-conv shape the network launches at a frame group of 4 and at the
batch-8 backend call, time the WHOLE stage that contains it (encode / pair decode) under each tile configuration
(mslam_gemm_tile_override) and keep the best - isolated warm loops over one shape mispredict.
The shapes are discovered by a child process run with MSLAM_GEMM_LOG=1.  Prints one line per shape; `BEST` lines are
the candidates for the measured table in csrc/gemm.hip (re-check them against the noise before adopting)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

PLAN = [("enc", 4), ("dec", 4), ("dec", 8)]
if os.environ.get("MSLAM_TUNE_PLAN"):   # e.g. "enc:1 enc:2 enc:3 dec:3"
    PLAN = [(t.split(":")[0], int(t.split(":")[1])) for t in os.environ["MSLAM_TUNE_PLAN"].split()]
H, W = 384, 512
CFGS = [642, 643, 644, 1262, 1263, 1242, 1282, 1283, 2128, 2256]
dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
L = m.lib()


def stage(kind, B):
    if kind == "enc":
        img = torch.rand(B, 3, H, W, device=dev) * 2 - 1
        return lambda: model._encode_image(img)
    f = torch.randn(B, 768, mc.enc_dim, device=dev)
    return lambda: model.decode_pair(f, f, H, W)


if len(sys.argv) > 1 and sys.argv[1] == "--list":
    for kind, B in PLAN:
        torch.cuda.synchronize()
        print(f"mslam_stage {kind} {B}", file=sys.stderr, flush=True)
        stage(kind, B)()
        torch.cuda.synchronize()
    sys.exit(0)


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


child = subprocess.run([sys.executable, os.path.abspath(__file__), "--list"], env=dict(os.environ, MSLAM_GEMM_LOG="1"),
                       stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, check=True)
shapes, cur = {}, None
for ln in child.stderr.splitlines():
    p = ln.split()
    if p[:1] == ["mslam_stage"]:
        cur = (p[1], int(p[2]))
        shapes[cur] = []
    elif p[:1] == ["mslam_gemm_shape"] and cur is not None:
        shapes[cur].append((int(p[1][5:]), int(p[2]), int(p[3]), int(p[4])))
seen_before = set()
for kind, B in PLAN:
    fn = stage(kind, B)
    fn(); fn()
    for conv, M, N, K in shapes[(kind, B)]:
        if (conv, M, N, K) in seen_before or 2.0 * M * N * K < 2e9:   # tuned in an earlier stage / too small to matter
            continue
        seen_before.add((conv, M, N, K))
        Ms = -M if conv else M
        base = timeit(fn)
        best_cfg, best_t = 0, base
        for cfg in CFGS:
            L.mslam_gemm_tile_override(Ms, N, K, cfg)
            try:
                fn()
                t = timeit(fn)
            except Exception:
                t = 1e9
            if t < best_t * 0.99:
                best_cfg, best_t = cfg, t
        L.mslam_gemm_tile_override(Ms, N, K, best_cfg)
        print(f"{'KEEP' if best_cfg == 0 else 'BEST'} {kind}{B} conv={conv} {M} {N} {K} cfg {best_cfg} stage {base:.3f} -> {best_t:.3f} ms", flush=True)
    print(f"== {kind}{B} final {timeit(fn):.3f} ms", flush=True)
