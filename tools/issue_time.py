#!/usr/bin/env python3
"""Host issue time vs GPU time of the network calls (is the launch path or the GPU the limit?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
H, W = 384, 512
img = torch.rand(1, 3, H, W, device=dev) * 2 - 1
feat = model._encode_image(img)[0]
model.decode_pair(feat, feat, H, W)
f4 = feat.expand(4, -1, -1).contiguous()
model.decode_pair(f4, f4, H, W)
for name, fn in (("encode B=1", lambda: model._encode_image(img)), ("decode B=1", lambda: model.decode_pair(feat, feat, H, W)),
                 ("decode B=4", lambda: model.decode_pair(f4, f4, H, W))):
    iss, tot = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        iss.append(t1 - t0); tot.append(t2 - t0)
    print(f"{name}: host issue {1e3 * min(iss):.2f} ms, until done {1e3 * min(tot):.2f} ms")
