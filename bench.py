#!/usr/bin/env python3
"""Headline benchmark: SLAM frames/sec of the per-frame hot path (MASt3R infer + match + TSDF fuse + GN
solve) on a synthetic 512x384 RGB-D stream (BASELINE.json config 3), N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE frame through the hot path:
  tracking   encode(frame) -> asymmetric decode + heads vs the last keyframe -> iterative-projection match
             + fp16 descriptor refinement -> frame-to-keyframe Sim3 Gauss-Newton
  every --kf-every-th frame additionally (keyframe / backend):
             symmetric decode + heads of --edges-per-kf keyframe pairs (both directions, batched),
             matching of both directions, global Sim3 GN over the keyframe graph, global TSDF integration
             of 40 000 points + TSDF pose refinement (3 iterations x 2 000 samples)
The network runs on random-init ViT-L weights of the real architecture (no checkpoint offline); because
random weights give meaningless geometry, the match / GN / TSDF stages consume seeded synthetic pointmaps
of the same shapes (back-projected room depth) - every stage does its full work.

Multi-GPU (weak scaling): every rank tracks its own frame stream; the keyframe graph grows with N and its
directed edges are sharded across ranks with one all-reduce of the normal-equation blocks per GN
iteration; TSDF voxels are sharded by key hash, keyframe points are all-gathered.
Prints ONE JSON line on rank 0.
"""
import argparse
import queue
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H, W = 384, 512
GF_TRACK = 1514.1      # GFLOP per tracked frame: encode 523.05 + decoder 437.28 + 2 heads x 276.90
GF_EDGE = 1982.2       # GFLOP per symmetric keyframe edge: 2 x (decoder + 2 heads)
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--kf-every", type=int, default=8)
    ap.add_argument("--edges-per-kf", type=int, default=4)
    ap.add_argument("--graph-kfs", type=int, default=8, help="keyframes in the backend graph PER GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks on cuda:0 with gloo collectives (rehearses the N>1 code path on a one-GPU box)")
    ap.add_argument("--graphs", action="store_true", help="replay the network as captured HIP graphs (default: eager)")
    ap.add_argument("--no-backend-thread", action="store_true",
                    help="run the keyframe backend inline in the tracking loop instead of on its own thread + stream")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="encode each frame inside its own step instead of one frame ahead on a second stream")
    ap.add_argument("--frame-group", type=int, default=4,
                    help="frames whose network stages run in one batch call: the encoder runs this many frames ahead, the "
                         "pair decode speculates that the keyframe stays (discarded and redone after a keyframe change); "
                         "matching and tracking stay strictly per frame.  1 = every stage one frame at a time")
    ap.add_argument("--depth-scale", type=float, default=1.0, help="debug: <1 shrinks the network depth")
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:   # rehearsal of the multi-rank path on one card: every rank on cuda:0, gloo collectives
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, world, torch.device("cuda", local if world > 1 else 0)


def barrier(world):
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


class BackendWorker(threading.Thread):
    """The reference runs the backend (symmetric edge inference, global GN, TSDF) in a process of its own
    beside the tracking frontend (main.py:73-163, tsdf_refine.py / global_manager.py threads); here it is a
    host thread with its own HIP stream.  Every queued task is finished before the clock stops."""

    def __init__(self, dev):
        super().__init__(daemon=True)
        self.q = queue.Queue()
        self.dev = dev
        self.error = None
        self.start()

    def run(self):
        torch.cuda.set_device(self.dev)   # the current device is per host thread
        stream = torch.cuda.Stream(device=self.dev)
        with torch.cuda.stream(stream):
            while True:
                task = self.q.get()
                try:
                    if task is None:
                        return
                    if self.error is None:
                        task()
                except Exception as e:  # surfaced by drain()
                    self.error = e
                finally:
                    self.q.task_done()

    def drain(self):
        self.q.join()
        if self.error is not None:
            raise self.error


class Pipeline:
    """Owns the model, the synthetic pools and one step of the hot path."""

    def __init__(self, args, rank, world, dev):
        from mast3r_slam import synthetic
        from mast3r_slam.config import config
        from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict
        from mast3r_slam.tracker import FrameTracker
        from mast3r_slam.tsdf import TSDFPoseOptimizer, TSDFVolume

        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        self.cfg = config
        ds = args.depth_scale
        mc = Mast3rConfig(enc_depth=max(1, round(24 * ds)), dec_depth=12)
        self.flop_scale = (523.05 * mc.enc_depth / 24 + 437.28 + 2 * 276.90) / GF_TRACK
        sd = random_state_dict(mc, seed=0)
        self.model = Mast3rHIP(sd, mc, device=dev, use_graphs=args.graphs)
        del sd
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        # RGB frame pool (ImgNorm range) - each rank its own stream segment
        self.frames = [t(synthetic.render_rgb(synthetic.camera_pose(8 * (rank * 100 + k)), H, W))[None] for k in range(4)]
        ts = torch.tensor([[H, W]])
        self.kf_feat = self.model._encode_image(self.frames[0], ts)[0]
        # geometry pool for matching + tracking: frame i vs keyframe j
        self.pairs = []
        for k in range(2):
            pr = synthetic.make_pair(8 * k + 3, 8 * k, h=H, w=W, seed=rank)
            Tf, Tk = synthetic.camera_pose(8 * k + 3), synthetic.camera_pose(8 * k)
            Xk = synthetic.render_pointmap(Tk, H, W).reshape(-1, 3).astype(np.float32)
            rng = np.random.default_rng(k)
            xi = rng.normal(0, 0.01, 7)
            Tf_noisy = Tf.copy(); Tf_noisy[:3] += xi[:3]
            self.pairs.append(dict(
                X11=t(pr["X11"])[None], X21=t(pr["X21"])[None], D11=t(pr["D11"])[None], D21=t(pr["D21"])[None],
                Xf=t(pr["X11"].reshape(-1, 3)), Xk=t(Xk), Qk=t(np.sqrt(pr["Q11"] * pr["Q21"]).reshape(-1, 1)),
                T_WCf=t(Tf_noisy.astype(np.float32)).reshape(1, 8), T_WCk=t(Tk.astype(np.float32)).reshape(1, 8)))
        self.tracker = FrameTracker(self.model, None, dev)
        # backend: symmetric edge batch features (reuse encoded keyframe features), graph, TSDF
        E = args.edges_per_kf
        self.feat_i = self.kf_feat.expand(E, -1, -1).contiguous()
        self.feat_j = self.model._encode_image(self.frames[1], ts)[0].expand(E, -1, -1).contiguous()
        self.feat_ij = torch.cat((self.feat_i, self.feat_j))
        self.feat_ji = torch.cat((self.feat_j, self.feat_i))
        g = synthetic.make_graph(n_kf=args.graph_kfs * world, h=H, w=W, seed=11, stride=4, extra_edges=2, pose_noise=0.01)
        self.graph = {k: t(v) for k, v in g.items() if isinstance(v, np.ndarray)}
        self.vol = TSDFVolume(0.03, 0.12, capacity=1 << 22, device=dev, shard_id=rank, num_shards=world)
        self.tsdf_opt = TSDFPoseOptimizer(self.vol, None, dict(config["tsdf_global"]), False, dev)
        Tk = synthetic.camera_pose(8 * rank)
        Xw = synthetic.sim3_act(Tk, synthetic.render_pointmap(Tk, H, W).reshape(-1, 3))
        rng = np.random.default_rng(rank)
        sel = rng.permutation(H * W)[:40000]
        self.tsdf_pts = t(Xw[sel].astype(np.float32))
        self.tsdf_conf = t(rng.uniform(0.5, 3.0, 40000))
        self.tsdf_org = t(Tk[:3].astype(np.float32))
        self.tsdf_cam_pts = t(synthetic.render_pointmap(Tk, H, W).reshape(-1, 3)[sel[:2000]].astype(np.float32))
        self.tsdf_cam_conf = t(rng.uniform(0.5, 3.0, 2000).astype(np.float32))
        self.tsdf_pose = t(Tk.astype(np.float32)).reshape(1, 8)
        # local (camera-side) TSDF refine: max_rois_per_kf 32x32-pixel blocks of the newest keyframe
        from lietorch_hip import Sim3
        from mast3r_slam.frame import Frame, KeyframeStore
        from mast3r_slam.tsdf_refine import PatchBlock, TSDFRefiner

        Xc = (synthetic.render_pointmap(Tk, H, W).reshape(-1, 3) + rng.normal(0, 0.003, (H * W, 3))).astype(np.float32)
        kf = Frame(0, self.frames[0], torch.tensor([[H, W]]), torch.tensor([[H, W]]), None,
                   Sim3(t(Tk.astype(np.float32)).reshape(1, 8)), t(Xc), t(rng.uniform(0.3, 1.0, (H * W, 1)).astype(np.float32)))
        kf.N = 1
        store = KeyframeStore()
        store.append(kf)
        self.refiner = TSDFRefiner(dict(config["tsdf_refine"]), store, None, dev)
        self.refine_C0 = kf.C.clone()
        self.refine_blocks = []
        for b in range(int(config["tsdf_refine"]["max_rois_per_kf"])):
            y0, x0 = 64 + 96 * b, 96 + 128 * b
            m = torch.zeros(H, W, dtype=torch.bool, device=dev)
            m[y0:y0 + 32, x0:x0 + 32] = True
            self.refine_blocks.append(PatchBlock(0, b, [], m.reshape(-1), 1.0, 1.0))
        # frontend pipeline: the encoder of frame f+1 runs on its own stream beside decode/match/track of frame f
        # multi-GPU: the backend thread is the only issuer of collectives while the clock runs (same order on every
        # rank); the main thread's barrier / all-reduce come after drain()
        self.worker = None if args.no_backend_thread else BackendWorker(dev)
        self.enc_stream = torch.cuda.Stream(device=dev)
        self.B = 1 if args.no_pipeline else max(1, args.frame_group)
        self.spec_waste = self.B // 2            # ceil((B-1)/2): expected frames decoded in vain per keyframe change
        self.t = 0                               # position in the stream (the step index restarts per phase)
        self.enc, self.enc_hi = {}, 0            # encoder batches in flight / done: first frame -> (feat, event)
        self.dec_hi, self.dec_epoch, self.kf_epoch = 0, -1, 0
        self.void_rows = 0                       # rows of the next group call that stand for frames decoded in vain
        self.kf_feat_b = {n: self.kf_feat.expand(n, -1, -1).contiguous() for n in range(1, self.B + self.spec_waste + 1)}
        self.img_b = {}
        self.net_ms = 0.0
        self.net_calls = 0
        self.timing = False

    def _net(self, fn):
        """Run a network stage; when timing, bracket it with events on the launch stream."""
        if not self.timing:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # local: two host threads
        e0.record()
        out = fn()
        e1.record()
        self._pending.append((e0, e1))
        return out

    def step(self, f):
        from lietorch_hip import Sim3
        from mast3r_slam import matching
        import mast3r_slam_backends as be

        a = self.args
        c = self.cfg
        pr = self.pairs[f % len(self.pairs)]
        t, B = self.t, self.B
        self.t += 1
        # ---- tracking ---------------------------------------------------------------------------
        main = torch.cuda.current_stream(self.dev)
        if t >= self.dec_hi or self.dec_epoch != self.kf_epoch:
            # frames t .. t+B-1 against the current keyframe in one call; their features were encoded >= B steps ago
            # on the encoder stream, which now runs the NEXT groups beside this one's decode / match / track
            self._encode_ahead(t)
            feats = self._group_feats(t, B, main)
            if self.void_rows:   # see the keyframe branch below
                feats = torch.cat((feats, feats[:self.void_rows]))
                self.void_rows = 0
            self._net(lambda: self.model.decode_pair(feats, self.kf_feat_b[feats.shape[0]], H, W))
            self.dec_hi, self.dec_epoch = t + B, self.kf_epoch
        idx, valid = matching.match(pr["X11"], pr["X21"], pr["D11"], pr["D21"])
        self.tracker.opt_pose_ray_dist_sim3(pr["Xf"], pr["Xk"], Sim3(pr["T_WCf"]), Sim3(pr["T_WCk"]), pr["Qk"],
                                            valid[0], idx=idx[0])
        # ---- keyframe / backend -------------------------------------------------------------------
        if f % a.kf_every == 0:
            # frame f became a keyframe: what was decoded ahead against the old one is void and the next group starts
            # right behind it.  With a fixed keyframe period the groups realign behind every keyframe and nothing would
            # ever be decoded in vain, so the EXPECTED loss for a keyframe at a random position of its group,
            # ceil((B-1)/2) rows of a group call, is charged explicitly: the next group call carries that many extra rows
            self.kf_epoch += 1
            self.void_rows = self.spec_waste
            if self.worker is not None:
                self.worker.q.put(self.backend)
            else:
                self.backend()

    def ahead(self):
        """(frames encoded, frames decoded) beyond the next frame to track."""
        dec = max(0, self.dec_hi - self.t) if self.dec_epoch == self.kf_epoch else 0
        return self.enc_hi - self.t, dec

    def settle(self, ahead0):
        """Called before the clock stops.  A timed region that starts with more frames already encoded / decoded ahead
        than it leaves behind for the next one (--steps not a multiple of the group size) has done less than --steps
        frames of network work: the difference is run here, inside the timed region, and discarded."""
        e0, d0 = ahead0
        e1, d1 = self.ahead()
        if e0 > e1:
            img = torch.cat([self.frames[k % len(self.frames)] for k in range(e0 - e1)])
            with torch.cuda.stream(torch.cuda.current_stream(self.dev) if self.args.no_pipeline else self.enc_stream):
                self._net(lambda: self.model._encode_image(img))
        if d0 > d1:
            n = d0 - d1
            self._net(lambda: self.model.decode_pair(self.kf_feat_b[n], self.kf_feat_b[n], H, W))

    def _encode_ahead(self, t):
        """Keep the encoder 2 groups ahead of frame t (batches of B frames, aligned at multiples of B)."""
        B = self.B
        while self.enc_hi < t + (B if self.args.no_pipeline else 2 * B):
            s0 = self.enc_hi
            key = s0 % len(self.frames)
            if key not in self.img_b:
                self.img_b[key] = torch.cat([self.frames[(s0 + k) % len(self.frames)] for k in range(B)])
            with torch.cuda.stream(torch.cuda.current_stream(self.dev) if self.args.no_pipeline else self.enc_stream):
                feat = self._net(lambda: self.model._encode_image(self.img_b[key])[0])
                ev = torch.cuda.Event()
                ev.record()
            self.enc[s0] = (feat, ev)
            self.enc_hi = s0 + B

    def _group_feats(self, t, n, main):
        B, parts, k = self.B, [], t
        while k < t + n:
            s0 = (k // B) * B
            feat, ev = self.enc[s0]
            main.wait_event(ev)
            feat.record_stream(main)
            e = min(t + n, s0 + B)
            parts.append(feat[k - s0:e - s0])
            k = e
        for s0 in [q for q in self.enc if q + B <= t]:
            del self.enc[s0]
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def backend(self):
        from lietorch_hip import Sim3
        from mast3r_slam import matching
        import mast3r_slam_backends as be

        a, c = self.args, self.cfg
        # symmetric inference of the E edges: both directions in one call of batch 2E (mast3r_decode_symmetric_batch)
        self._net(lambda: self.model.decode_pair(self.feat_ij, self.feat_ji, H, W))
        E = a.edges_per_kf
        X11 = torch.cat([p["X11"] for p in self.pairs] * E)[: 2 * E]
        X21 = torch.cat([p["X21"] for p in self.pairs] * E)[: 2 * E]
        D11 = torch.cat([p["D11"] for p in self.pairs] * E)[: 2 * E]
        D21 = torch.cat([p["D21"] for p in self.pairs] * E)[: 2 * E]
        matching.match(X11, X21, D11, D21)
        g = self.graph
        lc = c["local_opt"]
        Twc = g["Twc"].clone()
        if self.world > 1:
            from mast3r_slam.global_opt import gauss_newton_sharded

            gauss_newton_sharded("rays", Twc, g["Xs"], g["Cs"], None, g["ii"], g["jj"], g["idx_ii2jj"],
                                 g["valid_match"], g["Q"], lc)
        else:
            be.gauss_newton_rays(Twc, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx_ii2jj"], g["valid_match"], g["Q"],
                                 lc["sigma_ray"], lc["sigma_dist"], lc["C_conf"], lc["Q_conf"], lc["max_iters"],
                                 lc["delta_norm"])
        pts, conf, org = self.tsdf_pts, self.tsdf_conf, self.tsdf_org
        if self.world > 1:
            import torch.distributed as dist

            gp = [torch.empty_like(pts) for _ in range(self.world)]
            gc = [torch.empty_like(conf) for _ in range(self.world)]
            go = [torch.empty_like(org) for _ in range(self.world)]
            dist.all_gather(gp, pts); dist.all_gather(gc, conf); dist.all_gather(go, org)
            for r in range(self.world):
                self.vol.integrate(gp[r], gc[r], go[r], return_fused=False)
        else:
            self.vol.integrate(pts, conf, org, return_fused=False)
        self.tsdf_opt.refine_pose(Sim3(self.tsdf_pose), self.tsdf_cam_pts, self.tsdf_cam_conf, iterations=3)
        self.refiner.keyframes[0].C.copy_(self.refine_C0)
        for blk in self.refine_blocks:
            self.refiner.refine_block(blk)

    def network_probe(self, frames=8):
        """Kernel-quality figure for the roofline object: the network stages of `frames` tracked frames (in the
        pipeline's groups of B) and one keyframe batch run back to back WITHOUT the frontend overlap, event-timed on
        their stream.  Returns (GFLOP, ms)."""
        a, B = self.args, self.B
        ts = torch.tensor([[H, W]])
        groups = max(1, frames // B)
        img = torch.cat([self.frames[k % len(self.frames)] for k in range(B)])
        evs = []
        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = fn(); e1.record()
            evs.append((e0, e1))
            return out
        for rep in range(2):   # pass 0 untimed: this stream's arenas are allocated on first use
            evs.clear()
            torch.cuda.synchronize()
            for k in range(groups if rep else 1):
                feat = timed(lambda: self.model._encode_image(img, ts)[0])
                timed(lambda: self.model.decode_pair(feat, self.kf_feat_b[B], H, W))
            timed(lambda: self.model.decode_pair(self.feat_ij, self.feat_ji, H, W))
            torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for e0, e1 in evs)
        return groups * B * self.flop_scale * GF_TRACK + a.edges_per_kf * GF_EDGE, ms

    def dominant_kernel_probe(self, iters=50):
        """The kernel with the largest share of the step (profiles/r01_bench_kernel_stats.csv): the bf16 GEMM on the
        encoder's fc1 shape at the frame group's row count (768 B x 4096 x 1024, GELU) - the 256x256-tile, 16-wave
        instantiation from B = 3 up, the 64x64-tile one at B = 1 - launched through the C ABI, event-timed on its stream."""
        import mslam_hip as m

        M, N, K = 768 * self.B, 4096, 1024
        tiles256 = ((M + 255) // 256) * ((N + 255) // 256)
        name = ("gemm_bf16_kernel<4,4,2,2,2,false> (256x256 tile, 16 waves, LDS-DMA ring 2)" if tiles256 >= 128 else
                "gemm_bf16_kernel<2,4,2,1,2,false> (128x128 tile, 8 waves, LDS-DMA ring 2)" if tiles256 * 4 >= 300 else
                "gemm_bf16_kernel<2,2,1,1,2,false> (64x64 tile, LDS-DMA ring 2)")
        A = torch.randn(M, K, device=self.dev).to(torch.bfloat16)
        Wt = (torch.randn(N, K, device=self.dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=self.dev)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=self.dev)
        L = m.lib()
        call = lambda: L.mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), 0, m.ptr(out), M, N, K, 1, 1, m.stream_ptr())
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / iters
        return {"name": name, "shape": [M, N, K],
                "gflop_per_launch": 2e-9 * M * N * K, "us_per_launch": us, "tflops": 2e-6 * M * N * K / us}

    def gflop_per_step_avg(self):
        a = self.args
        return self.flop_scale * GF_TRACK + GF_EDGE * a.edges_per_kf / a.kf_every


def cpu_baseline(args):
    """The oracle (kind = "port") timed on the host cores for a BOUNDED sample of the same workload:
    one tracked frame (torch-CPU fp32 network restatement + C matching + numpy tracking GN) plus one
    keyframe's backend with ONE symmetric edge direction of network (scaled to --edges-per-kf), one GN
    iteration of the C restatement on a 4-keyframe graph (scaled) and the C TSDF on 40 000 points."""
    import oracle
    from oracle import mast3r_ref as R, matching_py, tracker_py
    from mast3r_slam import synthetic
    from mast3r_slam.config import config

    def note(msg):
        print(f"[cpu_baseline] {msg}", file=sys.stderr, flush=True)

    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, 16))   # a 1-GPU box grants a 16-core share; more threads only oversubscribe it
    os.environ["OMP_NUM_THREADS"] = str(cores)   # C oracle (OpenMP), read when liboracle.so is first loaded
    torch.set_num_threads(cores)
    note(f"{cores} cores; building random fp32 weights")
    cfg = R.Mast3rConfig()
    sd = R.init_state_dict(cfg, seed=0)
    img = torch.from_numpy(synthetic.render_rgb(synthetic.camera_pose(0), H, W))[None]
    note("network (torch CPU fp32): encoder")
    t0 = time.perf_counter()
    with torch.inference_mode():
        f1, p1 = R.encode_image(sd, cfg, img)
        t_enc = time.perf_counter() - t0
        note(f"encoder {t_enc:.1f}s; decoder")
        t0 = time.perf_counter()
        d1, d2 = R.decoder(sd, cfg, f1, p1, f1, p1)
        t_dec = time.perf_counter() - t0
        note(f"decoder {t_dec:.1f}s; head")
        t0 = time.perf_counter()
        R.downstream_head(sd, cfg, 1, d1, H, W)
        t_head = time.perf_counter() - t0
        note(f"head {t_head:.1f}s; matching")
    t_net_frame = t_enc + t_dec + 2 * t_head
    t_net_edge = 2 * (t_dec + 2 * t_head)
    del sd
    pr = synthetic.make_pair(3, 0, h=H, w=W, seed=0)
    t0 = time.perf_counter()
    rays, pts, p0 = matching_py.prep_for_iter_proj(pr["X11"][None], pr["X21"][None])
    mc = config["matching"]
    p, conv = oracle.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"])
    p1i, v = matching_py.occlusion_and_trunc(pr["X11"][None], pr["X21"][None], p, conv, mc["dist_thresh"])
    p1i = oracle.refine_matches(pr["D11"][None].astype(np.float16), pr["D21"].reshape(1, H * W, -1).astype(np.float16),
                                p1i, mc["radius"], mc["dilation_max"])
    t_match = time.perf_counter() - t0
    note(f"matching {t_match:.2f}s; tracking GN")
    idx = matching_py.pixel_to_lin(p1i, W)[0]
    Tk = synthetic.camera_pose(0)
    Xk = synthetic.render_pointmap(Tk, H, W).reshape(-1, 3).astype(np.float32)
    t0 = time.perf_counter()
    tracker_py.track(False, pr["X11"].reshape(-1, 3)[idx], Xk, synthetic.camera_pose(3).astype(np.float32),
                     Tk.astype(np.float32), np.sqrt(pr["Q11"] * pr["Q21"]).reshape(-1), v[0], dict(config["tracking"]))
    t_track = time.perf_counter() - t0
    note(f"tracking GN {t_track:.2f}s; backend GN")
    g = synthetic.make_graph(n_kf=4, h=H, w=W, seed=11, stride=4, extra_edges=1, pose_noise=0.01)
    lc = config["local_opt"]
    t0 = time.perf_counter()
    oracle.gauss_newton("rays", g["Twc"], g["Xs"], g["Cs"], None, g["ii"], g["jj"], g["idx_ii2jj"], g["valid_match"],
                        g["Q"], lc["sigma_ray"], lc["sigma_dist"], lc["C_conf"], lc["Q_conf"], 1, lc["delta_norm"])
    e_small = len(g["ii"])
    t_gn_edge_iter = (time.perf_counter() - t0) / e_small          # per directed edge per iteration (8 OpenMP threads)
    n_edges_bench = 2 * ((args.graph_kfs - 1) + 2 * (args.graph_kfs - 2))
    t_gn = t_gn_edge_iter * n_edges_bench * lc["max_iters"]
    note(f"GN {t_gn_edge_iter:.3f}s per edge-iteration; TSDF")
    Xw = synthetic.sim3_act(Tk, Xk.astype(np.float64))
    sel = np.random.default_rng(0).permutation(H * W)[:40000]
    vol = oracle.TSDFVolume(0.03, 0.12)
    t0 = time.perf_counter()
    vol.integrate(Xw[sel].astype(np.float32), np.full(40000, 2.0), Tk[:3].astype(np.float32))
    t_tsdf = time.perf_counter() - t0
    per_frame = t_net_frame + t_match + t_track
    per_kf = args.edges_per_kf * t_net_edge + 2 * args.edges_per_kf * t_match + t_gn + t_tsdf
    sec_per_frame = per_frame + per_kf / args.kf_every
    return dict(value=1.0 / sec_per_frame, unit="frames/s", cores=cores, kind="port",
                sample=("1 tracked frame (torch-CPU fp32 network %.1fs, C matching %.2fs, numpy tracking GN %.2fs) + "
                        "1 keyframe backend extrapolated from 1 decoder+heads pass, 1 GN iteration on %d edges, "
                        "40k-point TSDF %.2fs" % (t_net_frame, t_match, t_track, e_small, t_tsdf)))


def main():
    args = parse()
    rank, world, dev = dist_setup(args)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    import mslam_hip

    mslam_hip.check(mslam_hip.lib().mslam_device_check(), "device_check")
    pipe = Pipeline(args, rank, world, dev)
    # construction-time priming (not a warm-up step of the contract): one pass so that every arena, stream, fork
    # context and kernel attribute the pipeline allocates lazily exists before step 0, whatever --warmup is
    pipe.step(0)
    if pipe.worker is not None:
        pipe.worker.drain()
    barrier(world)
    for f in range(args.warmup):
        pipe.step(f)
    if pipe.worker is not None:
        pipe.worker.drain()
    barrier(world)
    pipe.timing = True
    pipe._pending = []
    ahead0 = pipe.ahead()
    t0 = time.perf_counter()
    for f in range(args.steps):
        pipe.step(f)
    pipe.settle(ahead0)
    t_enqueued = time.perf_counter() - t0   # host time to ISSUE the frontend work (no synchronisation inside)
    if pipe.worker is not None:
        pipe.worker.drain()   # every queued keyframe task has been issued ...
    barrier(world)            # ... and (device-wide synchronise inside) has finished
    elapsed = time.perf_counter() - t0
    net_ms = sum(a.elapsed_time(b) for a, b in pipe._pending)
    pipe.timing = False
    probe_gflop, probe_ms = pipe.network_probe()
    dom = pipe.dominant_kernel_probe()
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(f"[bench] frontend issue time {1e3 * t_enqueued / args.steps:.2f} ms/step of {1e3 * elapsed / args.steps:.2f} ms/step",
              file=sys.stderr)
        fps = args.steps * world / elapsed
        kf_steps = len([f for f in range(args.steps) if f % args.kf_every == 0])
        gflop_total = args.steps * pipe.flop_scale * GF_TRACK + kf_steps * args.edges_per_kf * GF_EDGE
        achieved = probe_gflop / max(probe_ms, 1e-9)  # GFLOP / ms = TFLOP/s
        out = {
            "metric": "SLAM frames/sec (infer+match+TSDF+GN) @512x384", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "synthetic 512x384 RGB-D stream, tracked frame every step + keyframe backend "
                                   f"every {args.kf_every} frames ({args.edges_per_kf} symmetric edges, "
                                   f"{args.graph_kfs * world}-keyframe GN graph, 40k-point TSDF fuse)",
                       "weights": "random-init ViT-L/12+12 MASt3R architecture (no checkpoint offline)",
                       "backend": "inline" if pipe.worker is None else "own host thread + stream (as the reference's backend process)",
                       "frontend": "eager launches" + (", HIP graphs" if args.graphs else "") +
                                   ("" if args.no_pipeline else ", encoder runs ahead on a second stream") +
                                   (f", network stages in groups of {pipe.B} frames (encoder ahead; pair decode speculative on the "
                                    f"keyframe, {pipe.spec_waste} discarded rows charged per keyframe); "
                                    "matching + tracking per frame" if pipe.B > 1 else ""),
                       "frame_group": pipe.B,
                       "parallelism": f"streams x{world}, GN edges + TSDF voxels sharded"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS,
                         # fabric-side bytes of ONE tracked frame's network pass (encode + decode), rocprofv3 --pmc
                         # FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE in separate passes: profiles/r01_pmc_hbm_traffic.json
                         # (measured for groups of 1 and of 4 frames; null for other group sizes)
                         "traffic": {1: 10.79e9, 4: 8.148e9}.get(pipe.B), "traffic_unit": "bytes per tracked-frame network pass (1.514 TFLOP)",
                         "dominant_kernel": dom,
                         "kernel": "gemm_bf16_kernel + attention_kernel (MASt3R forward: algorithmic GFLOP / event-timed "
                                   "stage ms, stages run back to back without the frontend overlap)",
                         "network_ms_per_step_overlapped": net_ms / args.steps,
                         "network_gflop_per_step": gflop_total / args.steps},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
