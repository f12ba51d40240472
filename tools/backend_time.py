#!/usr/bin/env python3
"""Isolated (nothing else on the GPU) event timing of the pieces of one tracked frame's match + track and of one
keyframe's backend, on the bench's own data: where the non-network time of a step goes."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import torch
import bench
from lietorch_hip import Sim3
from mast3r_slam import matching
import mast3r_slam_backends as be

sys.argv = [sys.argv[0], "--no-backend-thread"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
p = bench.Pipeline(args, 0, 1, dev)
H, W = bench.H, bench.W


def timeit(name, fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / n:8.3f} ms", flush=True)


pr = p.pairs[0]
timeit("match (1 pair)", lambda: matching.match(pr["X11"], pr["X21"], pr["D11"], pr["D21"]))
idx, valid = matching.match(pr["X11"], pr["X21"], pr["D11"], pr["D21"])
timeit("track GN (incl. verdict sync)", lambda: p.tracker.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], Sim3(pr["T_WCf"]), Sim3(pr["T_WCk"]), pr["Qk"], valid[0], idx=idx[0]))
print("  tracker iterations:", p.tracker.last_iters)
E = args.edges_per_kf
timeit(f"backend decode (batch {2 * E})", lambda: p.model.decode_pair(p.feat_ij, p.feat_ji, H, W), 5)
X11 = torch.cat([q["X11"] for q in p.pairs] * E)[: 2 * E]; X21 = torch.cat([q["X21"] for q in p.pairs] * E)[: 2 * E]
D11 = torch.cat([q["D11"] for q in p.pairs] * E)[: 2 * E]; D21 = torch.cat([q["D21"] for q in p.pairs] * E)[: 2 * E]
timeit(f"backend match (batch {2 * E})", lambda: matching.match(X11, X21, D11, D21), 5)
g, lc = p.graph, p.cfg["local_opt"]
def gn():
    Twc = g["Twc"].clone()
    be.gauss_newton_rays(Twc, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx_ii2jj"], g["valid_match"], g["Q"], lc["sigma_ray"],
                         lc["sigma_dist"], lc["C_conf"], lc["Q_conf"], lc["max_iters"], lc["delta_norm"])
timeit(f"backend GN ({len(g['ii'])} edges)", gn, 5)
timeit("TSDF integrate (40k points)", lambda: p.vol.integrate(p.tsdf_pts, p.tsdf_conf, p.tsdf_org, return_fused=False), 5)
timeit("TSDF pose refine (3 it)", lambda: p.tsdf_opt.refine_pose(Sim3(p.tsdf_pose), p.tsdf_cam_pts, p.tsdf_cam_conf, iterations=3), 5)
def refine():
    p.refiner.keyframes[0].C.copy_(p.refine_C0)
    for blk in p.refine_blocks:
        p.refiner.refine_block(blk)
timeit("local TSDF refine (3 blocks)", refine, 5)
timeit("whole backend()", p.backend, 5)
