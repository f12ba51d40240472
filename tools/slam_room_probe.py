#!/usr/bin/env python3
"""Functional probe of the product loop on the device-rendered room at full resolution WITHOUT the network (geometry
stand-in only): keyframe rate, relocalisations and trajectory error per frame.
    python tools/slam_room_probe.py [frames] [stride] [kf_thresh] [frame_group]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import numpy as np
import torch
from mast3r_slam import synthetic
from mast3r_slam.config import config
from mast3r_slam.frame import Frame
from mast3r_slam.slam_system import SlamSystem
from mast3r_slam.synthetic_gpu import PoseProximityRetriever, RoomGeometryModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 3
config["tracking"]["match_frac_thresh"] = float(sys.argv[3]) if len(sys.argv) > 3 else 0.72
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
H, W = 384, 512
dev = torch.device("cuda:0")
net = None
if os.environ.get("PROBE_NET"):
    from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict
    mc = Mast3rConfig(enc_depth=int(os.environ["PROBE_NET"]), dec_depth=12)
    net = Mast3rHIP(random_state_dict(mc, seed=0), mc, device=dev)
model = RoomGeometryModel(net, dev, H, W)
ret = PoseProximityRetriever(lambda fr: stride * int(fr.frame_id), max_dist=float(os.environ.get("PROBE_RDIST", 1.2)),
                             min_cos=float(os.environ.get("PROBE_RCOS", 0.75)))
if os.environ.get("PROBE_RETR") == "asmk":      # the product's retrieval database, random head + random codebook
    from mast3r_slam.retrieval_database import RetrievalDatabase, RetrievalWeights
    g = torch.Generator().manual_seed(1000)
    eye = torch.eye(1024, dtype=torch.float64)
    wh = lambda: (torch.zeros(1, 1024, dtype=torch.float64), eye + 0.02 * torch.randn(1024, 1024, generator=g, dtype=torch.float64))
    rw = RetrievalWeights(wh(), [(torch.randn(1024, 1024, generator=g) / 32.0, torch.zeros(1024))], wh(), nfeat=300, device=dev)
    ret = RetrievalDatabase(rw, torch.randn(65536, 1024, generator=g), device=dev)
    _upd = ret.update
    def _spy(frame, add_after_query, k, min_thresh=0.0):
        out = _upd(frame, add_after_query, k, min_thresh)
        sc = None if ret.last_scores is None else np.round(np.sort(ret.last_scores.cpu().numpy())[::-1][:4], 4).tolist()
        print(f"  retrieval: frame {int(frame.frame_id)} (path {stride * int(frame.frame_id)}) add={int(add_after_query)} -> db ids {out}, best scores {sc}", flush=True)
        return out
    ret.update = _spy
config["retrieval"]["k"] = int(os.environ.get("PROBE_K", 3))
tg = tr = qs = None
if os.environ.get("PROBE_TSDF"):
    from mast3r_slam.quality_async import SynchronousQualityService
    tg = dict(config["tsdf_global"], enabled=True, hash_capacity=1 << 22)
    if os.environ["PROBE_TSDF"] == "2":
        tg.update(pre_icp_iters=0, max_iterations=0)
    if os.environ.get("PROBE_NOQS"):
        qs = "none"
    tr = dict(config["tsdf_refine"], enabled=True)
    qs = None if qs == "none" else SynchronousQualityService(device=dev, lookup_both=True)
sys_ = SlamSystem(model, dev, retriever=ret, frame_group=B, backend=os.environ.get("PROBE_BACKEND", "inline"),
                  tsdf_global_cfg=tg, tsdf_refine_cfg=tr, quality_service=qs)
shp = torch.tensor([[H, W]])
k = stride * torch.arange(n, device=dev)
frames = []
for lo in range(0, n, 16):
    img = model.room.rgb(k[lo:lo + 16])
    frames += [Frame(lo + j, img[j:j + 1].clone(), shp, shp, None) for j in range(img.shape[0])]
keep = list(frames)
if os.environ.get("PROBE_TRACE"):
    res = []
    lo, hi = (int(v) for v in os.environ["PROBE_TRACE"].split(","))
    for i in range(n):
        res.append(sys_.step(frames[i]))
        if lo <= i <= hi:
            d = frames[i].T_WC.data.reshape(-1).cpu().numpy()
            kf = sys_.keyframes.last_keyframe()
            print(f"  step {i}: mode={res[-1]['mode'].name} new_kf={int(res[-1]['new_kf'])} reloc={int(res[-1]['try_reloc'])} iters={getattr(sys_.tracker, 'last_iters', -1)} "
                  f"T={np.round(d, 4).tolist()} kf_id={int(kf.frame_id)} kfT={np.round(kf.T_WC.data.reshape(-1).cpu().numpy(), 4).tolist()} "
                  f"kfX_absmax={float(kf.X_canon.abs().max()):.3f} kfC_mean={float(kf.get_average_conf().mean()):.3f}")
else:
    res = sys_.run(frames)
sys_.finish()
T0 = synthetic.camera_pose(0)
nkf = 0
for i, (f, r) in enumerate(zip(keep, res)):
    gt = synthetic.sim3_act(synthetic.sim3_inv(T0), synthetic.camera_pose(stride * i)[:3][None])[0]
    err = float(np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - gt))
    nkf += r["new_kf"]
    if r["try_reloc"] or r["mode"].name == "RELOC": print(f"  frame {i}: mode={r['mode'].name} try_reloc={int(r['try_reloc'])} err={err:.3f}")
    if os.environ.get("PROBE_SCALE") and i % 4 == 0: print(f"  frame {i:3d} err={err:.4f} scale={float(f.T_WC.data.reshape(-1)[7]):.4f} new_kf={int(r['new_kf'])}")
    if not os.environ.get("PROBE_QUIET"): print(f"frame {i:3d} mode={r['mode'].name:8s} new_kf={int(r['new_kf'])} reloc={int(r['try_reloc'])} err={err:.4f}")
print("max err over the last 10 frames:", max(float(np.linalg.norm(f.T_WC.data.reshape(-1)[:3].cpu().numpy() - synthetic.sim3_act(synthetic.sim3_inv(T0), synthetic.camera_pose(stride * i)[:3][None])[0])) for i, f in list(enumerate(keep))[-10:]))
fg = sys_.factor_graph
print("edges:", [(int(a), int(b)) for a, b in zip(fg.ii.tolist(), fg.jj.tolist())])
print("kf frame ids:", [int(sys_.keyframes[i].frame_id) for i in range(len(sys_.keyframes))])
print("relocs", sum(int(r["try_reloc"]) for r in res), "keyframes", nkf, "edges", int(sys_.factor_graph.ii.numel()), "stats", sys_.stats)
