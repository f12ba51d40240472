#!/usr/bin/env python3
"""K / N sweep of the bf16 GEMM at the tracked-frame row count (M = 768): separates the fixed cost of a
launch (prologue + epilogue + tail) from the per-K-tile cost.  python tools/gemm_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m
from bench_kernels import timeit, dev, L


def gemm(M, N, K, iters=100):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: L.mslam_gemm_bf16(m.ptr(A), m.ptr(W), m.ptr(bias), 0, m.ptr(out), M, N, K, 0, 1, m.stream_ptr()), iters)
    return us


if __name__ == "__main__":
    e = torch.empty(1, device=dev)
    us = timeit(lambda: e.zero_(), 200)
    print(f"torch zero_ (launch floor): {us:.2f} us")
    for M in (768, 1536, 3072):
        for N in (64, 768, 1024, 3072, 4096):
            row = []
            for K in (64, 256, 1024, 4096):
                row.append(gemm(M, N, K))
            print(f"M={M} N={N:5d}  K=64/256/1024/4096: " + "  ".join(f"{u:7.1f}" for u in row) + " us")
