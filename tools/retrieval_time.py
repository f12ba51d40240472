#!/usr/bin/env python3
"""Wall time of RetrievalDatabase.update (query + add, one keyframe) at the published sizes - 64k x 1024 codebook, 768
ViT-L tokens -> 300 local descriptors - with N images already in the database; the oracle (NumPy restatement of the
reference's python / Cython path) on the host beside it for the small N.
    python tools/retrieval_time.py 125 1250 [--cpu-upto 125]"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]
import numpy as np
import torch

from mast3r_slam.retrieval_database import RetrievalDatabase, RetrievalWeights

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="+")
ap.add_argument("--cpu-upto", type=int, default=125)
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
D = 1024
eye = torch.eye(D, dtype=torch.float64)
w = RetrievalWeights((torch.zeros(1, D, dtype=torch.float64), eye + 0.02 * torch.randn(D, D, generator=gen, dtype=torch.float64)),
                     [(torch.randn(D, D, generator=gen) / 32.0, torch.zeros(D))],
                     (torch.zeros(1, D, dtype=torch.float64), eye + 0.02 * torch.randn(D, D, generator=gen, dtype=torch.float64)),
                     nfeat=300, device=dev)
centroids = torch.randn(65536, D, generator=gen)
for N in sorted(args.sizes):
    db = RetrievalDatabase(w, centroids, device=dev)
    g2 = torch.Generator(device=dev).manual_seed(N)
    t_fill = time.perf_counter()
    for i in range(N):
        db.update(types.SimpleNamespace(feat=torch.randn(1, 768, D, device=dev, generator=g2)), True, 3, 0.0)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill
    probes = [types.SimpleNamespace(feat=torch.randn(1, 768, D, device=dev, generator=g2)) for _ in range(10)]
    db.update(probes[0], False, 3, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in probes:
        db.update(p, False, 3, 0.0)
    torch.cuda.synchronize()
    t_q = (time.perf_counter() - t0) / len(probes)
    # stage split with events
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record(); local = db.prep_features(probes[1].feat)[0]
    ev[1].record(); topk = db.quantize_custom(local, db.params["query_ivf"])
    ev[2].record(); sig, uniq = db._aggregate(local, topk)
    ev[3].record(); db.query(local)
    ev[4].record(); torch.cuda.synchronize()
    st = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    print(f"N={N:5d} images, {db._starts[-1]} entries: update(query only) {1e3 * t_q:7.2f} ms wall; fill {1e3 * t_fill / N:6.2f} ms/keyframe; "
          f"stages (ms): head {st[0]:.2f}, quantize {st[1]:.2f}, aggregate {st[2]:.2f}, whole query again {st[3]:.2f}", flush=True)
    if N <= args.cpu_upto:
        from oracle import asmk_py

        ref = asmk_py.RetrievalDatabase(None, centroids.numpy())
        ne = db._starts[-1]
        ref.ivf.words = db._e_word[:ne].cpu().numpy().astype(np.int64)
        ref.ivf.vecs = db._e_sig[:ne].cpu().numpy().view(np.uint32)
        ref.ivf.imids = np.repeat(np.arange(N), np.diff(db._starts))
        ref.ivf.norm_factor = np.diff(db._starts).astype(np.float64)
        ref.ivf.n_images = N
        ref.kf_counter = N
        loc = local.cpu().numpy()
        t0 = time.perf_counter()
        ref.update_local(loc, False, 3, 0.0)
        print(f"          oracle on the host (NumPy; quantise + aggregate + search of the same query): {time.perf_counter() - t0:6.2f} s", flush=True)
