#!/usr/bin/env python3
"""One batch call on one queue vs the same frames as k concurrent sub-batches on k streams (each with its own arena and
decoder fork): does inter-launch concurrency beat intra-launch batching at a frame group of 4 / 8?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
H, W = 384, 512
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run_split(fn_of_slice, B, k):
    main = torch.cuda.current_stream(dev)
    step = B // k
    evs = []
    for i in range(k):
        s = streams[i]
        s.wait_stream(main)
        with torch.cuda.stream(s):
            fn_of_slice(slice(i * step, (i + 1) * step))
            e = torch.cuda.Event(); e.record(); evs.append(e)
    for e in evs:
        main.wait_event(e)


def timeit(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (4, 8):
    img = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    feat = torch.randn(B, 768, mc.enc_dim, device=dev)
    enc = lambda sl: model._encode_image(img[sl])
    dec = lambda sl: model.decode_pair(feat[sl], feat[sl], H, W)
    for name, f in (("encode", enc), ("decode", dec)):
        line = [f"B={B} {name}: 1 queue {timeit(lambda: f(slice(0, B))):.3f} ms"]
        for k in (2, 4):
            line.append(f"{k} sub-batches {timeit(lambda: run_split(f, B, k)):.3f} ms")
        print("  ".join(line), flush=True)
