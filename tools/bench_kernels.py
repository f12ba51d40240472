#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written network kernels (not part of the test suite):
python tools/bench_kernels.py [shipped]  -> one line per shape with us/launch and TFLOP/s (shipped: the shapes of the
round-3 loop: encoder batches of 12, decode groups of 6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m

dev = torch.device("cuda:0")
L = m.lib()


def timeit(fn, iters=50):
    """us per call, GPU-side: the calls are captured once into a HIP graph and replayed, so the python /
    ctypes launch cost (several us) does not hide short kernels; also proves the entry points are
    capture-safe (no allocation, no synchronisation)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters  # us


def gemm(M, N, K, act=0, out_bf16=1):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev)
    us = timeit(lambda: L.mslam_gemm_bf16(m.ptr(A), m.ptr(W), m.ptr(bias), 0, m.ptr(out), M, N, K, act, out_bf16, m.stream_ptr()))
    print(f"gemm M={M:6d} N={N:5d} K={K:5d} act={act}: {us:8.1f} us  {2 * M * N * K / us * 1e-6:7.1f} TFLOP/s")


def conv(B, H, W, Cin, Cout, ks=3, stride=1):
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, ks * ks * Cin, device=dev) / (ks * ks * Cin) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    out = torch.empty((B, H // stride, W // stride, Cout), dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: L.mslam_conv2d_nhwc_bf16(m.ptr(x), m.ptr(w), m.ptr(bias), 0, m.ptr(out), B, H, W, Cin, Cout, ks, stride, 0, 0, m.stream_ptr()), 20)
    fl = 2 * B * (H // stride) * (W // stride) * Cout * ks * ks * Cin
    print(f"conv B={B} {H}x{W} {Cin}->{Cout} k{ks}s{stride}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s")


def attn(B, Hh, N):
    q = torch.randn(B, Hh, N, 64, device=dev).to(torch.bfloat16) * 0.125
    k = torch.randn(B, Hh, N, 64, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, Hh, 64, N, device=dev).to(torch.bfloat16)
    o = torch.empty(B, N, Hh * 64, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: L.mslam_attention_bf16(m.ptr(q), m.ptr(k), m.ptr(vt), m.ptr(o), B, Hh, N, N, m.stream_ptr()))
    print(f"attn B={B} H={Hh} N={N}: {us:8.1f} us  {4 * B * Hh * N * N * 64 / us * 1e-6:7.1f} TFLOP/s")


def ln(rows, D):
    x = torch.randn(rows, D, device=dev)
    w, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    ob = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: L.mslam_layernorm_f32(m.ptr(x), m.ptr(w), m.ptr(b), m.ptr(ob), 0, rows, D, 1e-6, m.stream_ptr()))
    print(f"layernorm {rows}x{D}: {us:8.1f} us  {rows * D * 6 / us * 1e-3:7.1f} GB/s")


if __name__ == "__main__" and os.path.basename(sys.argv[0]) == "bench_kernels.py" and "shipped" in sys.argv[1:]:
    # the shapes of the shipped loop (round 3): encoder on 12 frames (9 216 rows), decoder + heads on 6 pairs (4 608 rows per
    # side; both sides in one grouped launch inside the network - here one side alone), backend batch of 8 rows
    print("# encoder, 12 frames")
    for shp in [(9216, 3072, 1024), (9216, 1024, 1024), (9216, 1024, 4096)]:
        gemm(*shp)
    gemm(9216, 4096, 1024, act=1)
    attn(12, 16, 768)
    ln(9216, 1024)
    print("# decoder, 6 pairs (one side)")
    for shp in [(4608, 2304, 768), (4608, 768, 768), (4608, 1536, 768), (4608, 768, 3072)]:
        gemm(*shp)
    gemm(4608, 3072, 768, act=1)
    attn(12, 12, 768)
    ln(4608, 768)
    print("# heads, 6 images")
    conv(6, 384, 512, 128, 128)
    conv(6, 192, 256, 256, 128)
    conv(6, 96, 128, 256, 256)
    conv(6, 48, 64, 256, 256)
    conv(6, 24, 32, 768, 256)
    conv(6, 24, 32, 1024, 96, ks=1)
    gemm(4608, 7168, 1792)
    gemm(4608, 6400, 7168)
    sys.exit(0)

if __name__ == "__main__" and os.path.basename(sys.argv[0]) == "bench_kernels.py":
    for shp in [(768, 1024, 1024), (768, 3072, 1024), (768, 4096, 1024), (768, 1024, 4096), (768, 768, 768),
                (768, 2304, 768), (768, 3072, 768), (768, 768, 3072), (6144, 1024, 1024), (6144, 4096, 1024),
                (6144, 1024, 4096), (768, 7168, 1792), (768, 6400, 7168), (4096, 4096, 4096), (8192, 8192, 8192)]:
        gemm(*shp)
    gemm(768, 4096, 1024, act=1)
    conv(1, 384, 512, 128, 128)
    conv(1, 192, 256, 256, 128)
    conv(1, 192, 256, 256, 256)
    conv(1, 96, 128, 256, 256)
    conv(1, 24, 32, 768, 256)
    conv(1, 24, 32, 1024, 96, ks=1)
    attn(1, 16, 768); attn(2, 12, 768); attn(8, 16, 768)
    ln(768, 1024); ln(6144, 1024)
