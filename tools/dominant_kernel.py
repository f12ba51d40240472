#!/usr/bin/env python3
"""Launches only the kernel bench.py names as dominant (bf16 GEMM on the encoder's fc1 shape at the frame group's row
count: 256x256-tile instantiation at 3072 x 4096 x 1024 for the default group of 4; `dominant_kernel.py 1` gives the
64x64-tile one at 768 rows), GELU epilogue, so
that `rocprofv3 --kernel-trace --stats -- python3 tools/dominant_kernel.py` gives its average duration in isolation
(profiles/r01_dominant_kernel_stats.csv) next to the event-timed figure bench.py prints."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import mslam_hip as m

dev = torch.device("cuda:0")
M, N, K = 768 * (int(sys.argv[1]) if len(sys.argv) > 1 else 4), 4096, 1024
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
L = m.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        L.mslam_gemm_bf16(m.ptr(A), m.ptr(W), m.ptr(bias), 0, m.ptr(out), M, N, K, 1, 1, m.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
print(f"event-timed: {1e3 * e0.elapsed_time(e1) / 50:.2f} us per launch, {2e-6 * M * N * K / (1e3 * e0.elapsed_time(e1) / 50):.0f} TFLOP/s")
