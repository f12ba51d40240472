#!/usr/bin/env python3
"""A/B of tile configurations on given plain-GEMM shapes in ONE process, interleaved rounds (median and min of the
event-timed launch), with a correctness check against torch fp32 on the same bf16 operands.
    python tools/gemm_cfg_ab.py "3072,4096,1024,1" "6144,4096,1024,1" --cfgs 2256 2192 1282     (M,N,K,act)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m

ap = argparse.ArgumentParser()
ap.add_argument("shapes", nargs="+")
ap.add_argument("--cfgs", type=int, nargs="+", default=[0, 2256, 2192, 1282])
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-check", action="store_true", help="skip the correctness check (ablation builds: MSLAM_LIB=...)")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = m.lib()
for sh in a.shapes:
    M, N, K, act = (int(v) for v in sh.split(","))
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ref = A[:256].float() @ Wt.float().T + bias
    ref = torch.nn.functional.gelu(ref) if act == 1 else ref
    times = {c: [] for c in a.cfgs}
    for c in a.cfgs:
        L.mslam_gemm_tile_override(M, N, K, c)
        out.zero_()
        L.mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), 0, m.ptr(out), M, N, K, act, 1, m.stream_ptr())
        err = (out[:256].float() - ref).abs().max().item() / ref.abs().max().item()
        assert a.no_check or err < 2e-2, (c, err)
    for r in range(a.rounds):
        for c in a.cfgs:
            L.mslam_gemm_tile_override(M, N, K, c)
            for _ in range(3):
                L.mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), 0, m.ptr(out), M, N, K, act, 1, m.stream_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                L.mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), 0, m.ptr(out), M, N, K, act, 1, m.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            times[c].append(1e3 * e0.elapsed_time(e1) / a.iters)
    L.mslam_gemm_tile_override(M, N, K, 0)
    fl = 2e-6 * M * N * K
    print(f"{M}x{N}x{K} act={act}: " + "  ".join(
        f"cfg {c}: med {sorted(t)[len(t)//2]:.1f} us ({fl / sorted(t)[len(t)//2]:.0f} TF) min {min(t):.1f}" for c, t in times.items()), flush=True)
