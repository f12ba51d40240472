#!/usr/bin/env python3
"""Per-frame cost of the encoder and of the pair decode as a function of the number of frames in one call
(batch_time.py [iters]): what frame-group batching of the tracked path can buy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
H, W = 384, 512
ts = torch.tensor([[H, W]])


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (1, 2, 3, 4, 8):
    img = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    te = timeit(lambda: model._encode_image(img, ts))
    feat = model._encode_image(img, ts)[0]
    kf = feat[:1].expand(B, -1, -1).contiguous()
    td = timeit(lambda: model.decode_pair(feat, kf, H, W))
    print(f"B={B}: encode {te:.3f} ms ({te / B:.3f}/frame)  decode {td:.3f} ms ({td / B:.3f}/frame)", flush=True)
