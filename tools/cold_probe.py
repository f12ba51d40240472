#!/usr/bin/env python3
"""Why do the network's GEMMs run ~1.4x slower in situ than in a warm loop?  Times the encoder's qkv GEMM
(768 x 3072 x 1024) with (a) everything warm, (b) the activation freshly written by another kernel before every
launch (what a layer chain does), (c) weights rotated through 64 matrices = 400 MB (colder than the 256 MB
Infinity Cache), (d) both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"), os.path.join(ROOT, "tools")]
import torch
import mslam_hip as m
from bench_kernels import dev, L

M, N, K = 768, 3072, 1024
NW = 64
A0 = torch.randn(M, K, device=dev).to(torch.bfloat16)
A = A0.clone()
Ws = [(torch.randn(N, K, device=dev) / 32).to(torch.bfloat16) for _ in range(NW)]
bias = torch.randn(N, device=dev)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)


def run(cold_a, cold_w, gemm=True, iters=64, nw=NW):
    def body():
        for i in range(iters):
            if cold_a:
                A.copy_(A0)                      # a producer kernel rewrites the activation (lands in ITS XCDs' L2s)
            if gemm:
                W = Ws[i % nw] if cold_w else Ws[0]
                L.mslam_gemm_bf16(m.ptr(A), m.ptr(W), m.ptr(bias), 0, m.ptr(out), M, N, K, 0, 1, m.stream_ptr())
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


if __name__ == "__main__":
    copy_only = run(True, False, gemm=False)
    print(f"copy alone {copy_only:.1f} us")
    print(f"warm            {run(False, False):.1f} us")
    print(f"fresh A         {run(True, False) - copy_only:.1f} us (copy subtracted)")
    print(f"cold W          {run(False, True):.1f} us")
    for nw in (2, 4, 8, 16, 32):
        print(f"W rotating through {nw:2d} matrices ({nw * 6.3:.0f} MB): {run(False, True, nw=nw):.1f} us")
    print(f"fresh A, cold W {run(True, True) - copy_only:.1f} us (copy subtracted)")
