#!/usr/bin/env python3
"""The hand-written bf16 GEMM against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the network's main
shapes, plain epilogue (no bias / activation on either side), each alone on the GPU, event-timed.
    python tools/gemm_vs_blas.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m

dev = torch.device("cuda:0")
L = m.lib()
shapes = [(3072, 4096, 1024, "enc fc1, group of 4"), (3072, 1024, 4096, "enc fc2"), (3072, 3072, 1024, "enc qkv"),
          (3072, 1024, 1024, "enc proj"), (768, 4096, 1024, "enc fc1, 1 frame"), (768, 3072, 768, "dec fc1, 1 row"),
          (768, 768, 3072, "dec fc2"), (768, 2304, 768, "dec qkv"), (3072, 3072, 768, "dec fc1, 4 rows"),
          (8192, 8192, 8192, "large")]


def timed(fn, n):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"{'shape':>22s} {'note':>22s} | ours us  TF/s | library us  TF/s | ours/library")
for M, N, K, note in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    Wt = W.t()
    n = 10 if M >= 8192 else 50
    t_ours = timed(lambda: L.mslam_gemm_bf16(m.ptr(A), m.ptr(W), 0, 0, m.ptr(out), M, N, K, 0, 1, m.stream_ptr()), n)
    ref = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    t_lib = timed(lambda: torch.matmul(A, Wt, out=ref), n)
    err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
    gf = 2e-9 * M * N * K
    print(f"{M:6d}x{N:5d}x{K:5d} {note:>22s} | {t_ours:7.1f} {gf * 1e3 / t_ours:5.0f} | {t_lib:10.1f} {gf * 1e3 / t_lib:5.0f} | {t_lib / t_ours:5.2f}x  (max rel diff {err:.1e})",
          flush=True)
