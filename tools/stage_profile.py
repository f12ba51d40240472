#!/usr/bin/env python3
"""Runs ONE network stage repeatedly (for rocprofv3 --kernel-trace --stats): stage_profile.py enc|enc<B>|dec<B> [n]
(enc12 = the encoder on a batch of 12 frames, dec6 = the two-view decoder + heads on 6 pairs, ...)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict

stage, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
mc = Mast3rConfig()
model = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
H, W = 384, 512
img = torch.rand(1, 3, H, W, device=dev) * 2 - 1
feat = torch.randn(1, (H // 16) * (W // 16), mc.enc_dim, device=dev)   # values do not matter for time / traffic
B = int(stage[3:]) if len(stage) > 3 else 1
fb = feat.expand(B, -1, -1).contiguous()
imgB = img.expand(B, -1, -1, -1).contiguous()
for _ in range(n):
    if stage.startswith("enc"):
        model._encode_image(imgB)
    else:
        model.decode_pair(fb, fb, H, W)
torch.cuda.synchronize()
