import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import numpy as np, torch
from mast3r_slam import synthetic, matching
from mast3r_slam.synthetic_gpu import RoomRenderer
dev = torch.device("cuda:0")
for (H, W) in ((96, 128), (384, 512)):
    R = RoomRenderer(dev, H, W)
    for k in (3, 6, 12, 24):
        a, b = R.pair(torch.tensor([float(k)], device=dev), torch.tensor([0.0], device=dev), noise=0.002)
        idx, valid = matching.match(a["pts3d"], b["pts3d"], a["desc"], b["desc"])
        v = valid[0, :, 0]
        uniq = torch.unique(idx[0][v]).numel() / v.numel()
        pr = synthetic.make_pair(k, 0, h=H, w=W, seed=1, noise=0.002)
        t = lambda x: torch.from_numpy(x[None]).to(dev)
        idx2, valid2 = matching.match(t(pr["X11"]), t(pr["X21"]), t(pr["D11"]), t(pr["D21"]))
        v2 = valid2[0, :, 0]
        uniq2 = torch.unique(idx2[0][v2]).numel() / v2.numel()
        print(f"{H}x{W} k={k:2d}: device-render valid {v.float().mean():.3f} unique {uniq:.3f} | numpy make_pair valid {v2.float().mean():.3f} unique {uniq2:.3f}", flush=True)
