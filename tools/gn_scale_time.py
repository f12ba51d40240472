#!/usr/bin/env python3
"""Event timing of the global Gauss-Newton at the graph sizes BASELINE config 3 / 5 reach (384x512 pointmaps):
P keyframes, consecutive + 3 earlier edges each, both directions.  Device-generated data of the right shapes and
statistics (near-identity correspondences, ~60 % valid): timing only, parity lives in tests/test_gn_gpu.py.
    python tools/gn_scale_time.py 16 63 125 [--iters 10] [--hw 196608]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import numpy as np
import torch

import mast3r_slam_backends as be
import mslam_hip as m

ap = argparse.ArgumentParser()
ap.add_argument("poses", type=int, nargs="+")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--hw", type=int, default=384 * 512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--graph", choices=("random", "band"), default="random",
                help="random: consecutive + 3 random earlier keyframes (SURVEY 8d: no structure, the envelope is full); "
                     "band: consecutive + the 2 before + a loop closure to an early keyframe every 40th (a trajectory)")
args = ap.parse_args()
dev = torch.device("cuda:0")
HW = args.hw
for P in args.poses:
    rng = np.random.default_rng(P)
    und = [(k - 1, k) for k in range(1, P)]
    for k in range(2, P):
        if args.graph == "random":
            for a in rng.choice(k - 1, size=min(3, k - 1), replace=False):
                und.append((int(a), k))
        else:
            und += [(k - 2, k)] + ([(k - 3, k)] if k >= 3 else [])
            if k % 40 == 0:
                und.append((int(rng.integers(0, max(1, k // 3))), k))
    ii = torch.tensor([a for a, b in und] + [b for a, b in und], device=dev)
    jj = torch.tensor([b for a, b in und] + [a for a, b in und], device=dev)
    E = ii.numel()
    g = torch.Generator(device=dev).manual_seed(P)
    base = torch.randn(HW, 3, device=dev, generator=g) * 0.5 + torch.tensor([0.0, 0.0, 3.0], device=dev)
    Xs = (base[None] + 0.002 * torch.randn(P, HW, 3, device=dev, generator=g)).contiguous()
    Cs = torch.rand(P, HW, 1, device=dev, generator=g) * 2 + 1
    Twc = torch.zeros(P, 8, device=dev)
    Twc[:, 6] = 1
    Twc[:, 7] = 1
    Twc[1:, :3] = 0.003 * torch.randn(P - 1, 3, device=dev, generator=g)
    idx = torch.arange(HW, device=dev)[None].expand(E, HW).contiguous()
    valid = (torch.rand(E, HW, 1, device=dev, generator=g) < 0.6)
    Q = torch.rand(E, HW, 1, device=dev, generator=g) * 3 + 1.6
    ws_gb = m.lib().mslam_gn_workspace_bytes(P, E, HW, E) / 1e9

    def run():
        T = Twc.clone()
        be.gauss_newton_rays(T, Xs, Cs, ii, jj, idx, valid, Q, 0.003, 10.0, 0.0, 1.5, args.iters, 1e-8)
        return T

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        T = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    print(f"P={P:5d} E={E:6d} unknowns={7 * (P - 1):6d} workspace={ws_gb:7.2f} GB  GN {args.iters} it: {ms:9.2f} ms "
          f"({ms / args.iters:8.3f} ms/it)  finite={bool(torch.isfinite(T).all())}", flush=True)
    del Xs, Cs, idx, valid, Q
    torch.cuda.empty_cache()
