#!/usr/bin/env python3
"""Headline benchmark: SLAM frames/sec of the per-frame hot path (MASt3R infer + match + TSDF fuse + GN
solve) on a synthetic 512x384 RGB-D stream (BASELINE.json config 3), N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The timed region is the PRODUCT loop: `SlamSystem.run()` (mast3r_slam/slam_system.py = the reference's main.py frontend
loop + run_backend, one process) over frames of the procedural room, followed by the drain of its backend.
A "step" is one frame through that loop:
  tracking   encode(frame) -> asymmetric decode + heads vs the last keyframe -> iterative-projection match + fp16
             descriptor refinement -> frame-to-keyframe Sim3 Gauss-Newton -> keyframe decision (match fractions)
  per new keyframe (backend: three stages - graph | solve | fusion - on host threads + streams of their own, so that
             consecutive keyframe tasks overlap; the reference's backend is a process of its own):
             retrieval of <= 3 earlier keyframes + the consecutive one -> symmetric decode + heads of those pairs (both
             directions, one batch) -> matching of both directions -> global Sim3 GN over the WHOLE keyframe graph ->
             global TSDF: fuse the keyframe's 40 000 points, re-fuse / pose-refine queued keyframes -> local TSDF block
             refinement of the keyframe that left the sliding window
Nothing about the schedule is modelled: keyframes, speculative decode rows, graph and voxel-table growth are whatever the
loop decides.  The network runs on random-init ViT-L weights of the real architecture (no checkpoint offline); because
random weights give meaningless geometry, the model wrapper (mast3r_slam.synthetic_gpu.RoomGeometryModel) launches the
real network for every call and hands the room's geometry to the rest of the loop (its on-device rendering is extra
work inside the timed region).  Retrieval is the pose-proximity stand-in (the ASMK codebook is not available).

Graph size: config 3 is a 1 000-frame stream whose keyframe graph grows from 1 to ~125 keyframes.  `--steps 1000`
times exactly that (no pre-roll).  A short run (the default) first tracks `--preroll` frames untimed (default 500:
~63 keyframes, the mean of the schedule) so that the timed steps solve a graph of the mean size.

Multi-GPU.  `python bench.py --gpus N` starts its own N ranks (one process per GPU, before anything touches a GPU) unless
it already runs under torch.distributed.run.  Two things are measured at N > 1, one after the other:
  * `value` (weak scaling): every rank runs its own session (replicas: tracking is sequential in time, there is no
    data-path collective between independent streams);
  * `sharded_backend` (what BASELINE configs 4 / 5 and the north_star name): ONE session whose backend is sharded over the
    N ranks (mast3r_slam/shard.py) - rank 0 tracks and drives, keyframe-pair inference + matching are split over the
    ranks and all-gathered, the global GN accumulates each rank's edges and sums the normal-equation blocks with ONE
    all-reduce per iteration (RCCL over xGMI), the global TSDF's voxels live on their owner ranks.  `--mode` picks one.
    The replicas are measured first; the sharded session runs behind a watchdog (BENCH_SHARD_TIMEOUT, default 300 s) so
    that a failure there still leaves the line with `value` and `sharded_backend: {"error": ...}`.
Prints ONE JSON line on rank 0.
"""
import argparse
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
GF_ENC, GF_DEC, GF_HEAD = 523.05, 437.28, 276.90     # GFLOP: encoder per view, decoder per pair-direction, head per view
GF_TRACK = GF_ENC + GF_DEC + 2 * GF_HEAD             # 1514.1 per tracked frame
GF_EDGE = 2 * (GF_DEC + 2 * GF_HEAD)                 # 1982.2 per symmetric keyframe edge
PEAK_BF16_TFLOPS = 2500.0                            # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--preroll", type=int, default=-1,
                    help="frames tracked untimed before the warm-up so that the timed steps see the mean graph of the "
                         "1 000-frame schedule (-1: 500 for runs shorter than 500 steps, else 0)")
    ap.add_argument("--stride", type=int, default=3, help="camera-path steps per frame")
    ap.add_argument("--kf-thresh", type=float, default=0.5,
                    help="tracking.match_frac_thresh: a new keyframe when the match / unique fraction drops below it.  On this "
                         "camera path at 384x512 the unique-match fraction falls from 0.61 (3 path steps) to 0.48 (24 steps): "
                         "0.5 gives a keyframe every ~8 frames, the rate BASELINE config 3 names (the reference default 0.333 "
                         "would give one every ~40)")
    ap.add_argument("--retrieval-k", type=int, default=3)
    ap.add_argument("--retriever", choices=("pose", "asmk"), default="pose",
                    help="loop-closure proposals: 'pose' = the pose-proximity stand-in (what a trained retrieval model would "
                         "return on this scene); 'asmk' = the product's RetrievalDatabase (mast3r_slam/retrieval_database.py) "
                         "with a random retrieval head and a random 64k x 1024 codebook on the random-weight encoder's tokens")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-frames", action="store_true",
                    help="PCIe-inclusive variant (DESIGN.md; never the headline value): the RGB frames wait in pinned host "
                         "memory and every run() call first uploads its frames on the tracking stream")
    ap.add_argument("--mode", choices=("auto", "replicas", "shard-backend"), default="auto",
                    help="what is timed at N > 1: 'replicas' = N independent sessions (the headline value), 'shard-backend' = one "
                         "session whose backend is sharded over the N ranks, 'auto' = both, one after the other")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks on cuda:0 with gloo collectives (rehearses the N>1 code path on a one-GPU box)")
    ap.add_argument("--graphs", action="store_true", help="replay the network as captured HIP graphs (default: eager)")
    ap.add_argument("--no-backend-thread", action="store_true",
                    help="run the keyframe backend inline in the tracking loop (the reference's single_thread mode)")
    ap.add_argument("--frame-group", type=int, default=6,
                    help="frames whose network stages run in one batch call (SlamSystem frame groups)")
    ap.add_argument("--backend-stages", type=int, default=3,
                    help="threaded backend: stages of the keyframe task on threads / streams of their own - graph stage "
                         "(retrieval, pair inference, matching) | solve stage (global GN) | fusion stage (TSDF, local "
                         "refinement; 2 = together with the solve); 1 = one thread does the whole keyframe task")
    ap.add_argument("--solve-priority", type=int, default=0,
                    help="HIP stream priority of the backend's solve stage (-1 = high): a chain of short kernels and host reads")
    ap.add_argument("--encoder-group", type=int, default=12,
                    help="frames per look-ahead encoder call (0 = the frame group; every frame is encoded exactly once, so "
                         "a larger batch costs nothing but look-ahead)")
    ap.add_argument("--decode-ahead", type=int, default=0,
                    help="SlamSystem decode_ahead: issue the next group's pair decode on its own stream when at most this "
                         "many decoded frames are left (0: on the tracking stream when none is left)")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="SlamSystem pipeline_depth: frames whose matching + pose solve are enqueued before the oldest verdict is "
                         "read (0 = the default: frame-at-a-time loop, which measures faster end to end: DESIGN.md)")
    ap.add_argument("--encoder-priority", type=int, default=0,
                    help="HIP stream priority of the look-ahead encoder stream (-1 = high)")
    ap.add_argument("--backend-priority", type=int, default=0,
                    help="HIP stream priority of the backend thread's stream (-1 = high)")
    ap.add_argument("--tracking-priority", type=int, default=0,
                    help="run the tracking loop on a stream of this priority (-1 = high) instead of the default stream")
    ap.add_argument("--no-tsdf", action="store_true", help="debug: global + local TSDF off")
    ap.add_argument("--no-network", action="store_true", help="debug: geometry stand-in only, no network launches")
    ap.add_argument("--depth-scale", type=float, default=1.0, help="debug: <1 shrinks the encoder depth")
    return ap.parse_args()


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """The collective libraries print a banner (RCCL: version / hostname / library path, gloo: peer connections) on
    STDOUT when a communicator is created; the contract is ONE JSON line there, so file descriptor 1 points at stderr
    while the process group comes up."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with stdout_to_stderr():
            if args.share_gpu:   # rehearsal of the multi-rank path on one card: every rank on cuda:0, gloo collectives
                local = 0
                torch.cuda.set_device(0)
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            warm = torch.ones(1, device=torch.device("cuda", local))   # communicators are created by the first collective
            dist.all_reduce(warm)
            dist.barrier()
            torch.cuda.synchronize()
    else:
        torch.cuda.set_device(0)
    return rank, world, torch.device("cuda", local if world > 1 else 0)


def barrier(world):
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    torch.cuda.synchronize()


class Session:
    """One SLAM session of the product on the procedural room."""

    def __init__(self, args, rank, world, dev, total_frames, channel=None):
        from mast3r_slam.config import config
        from mast3r_slam.frame import Frame
        from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict
        from mast3r_slam.quality_async import SynchronousQualityService
        from mast3r_slam.slam_system import SlamSystem
        from mast3r_slam.synthetic_gpu import PoseProximityRetriever, RoomGeometryModel

        self.args, self.rank, self.dev = args, rank, dev
        config["tracking"]["match_frac_thresh"] = args.kf_thresh
        config["retrieval"]["k"] = args.retrieval_k
        self.n_path = 1000
        self.base = 37 * rank                      # every rank its own segment of the camera path
        self.net = None
        self.gf_enc = 0.0
        if not args.no_network:
            mc = Mast3rConfig(enc_depth=max(1, round(24 * args.depth_scale)), dec_depth=12)
            self.gf_enc = GF_ENC * mc.enc_depth / 24
            sd = random_state_dict(mc, seed=0)
            self.net = Mast3rHIP(sd, mc, device=dev, use_graphs=args.graphs)
            del sd
        self.model = RoomGeometryModel(self.net, dev, H, W, n_frames=self.n_path, seed=rank)
        path_index = lambda fr: self.base + args.stride * int(fr.frame_id)
        retriever = PoseProximityRetriever(path_index, self.n_path)
        if args.retriever == "asmk":
            from mast3r_slam.retrieval_database import RetrievalDatabase, RetrievalWeights

            g = torch.Generator().manual_seed(1000 + rank)
            eye = torch.eye(1024, dtype=torch.float64)
            wh = lambda: (torch.zeros(1, 1024, dtype=torch.float64), eye + 0.02 * torch.randn(1024, 1024, generator=g, dtype=torch.float64))
            rw = RetrievalWeights(wh(), [(torch.randn(1024, 1024, generator=g) / 32.0, torch.zeros(1024))], wh(), nfeat=300, device=dev)
            retriever = RetrievalDatabase(rw, torch.randn(65536, 1024, generator=g), device=dev)
        tg, tr = tsdf_cfgs(args)
        qs = SynchronousQualityService(device=dev, lookup_both=True) if tg is not None else None
        self.system = SlamSystem(self.model, dev, retriever=retriever, frame_group=max(1, args.frame_group), encoder_group=(args.encoder_group or None),
                                 tsdf_global_cfg=tg, tsdf_refine_cfg=tr, quality_service=qs, decode_ahead=args.decode_ahead,
                                 backend="inline" if args.no_backend_thread else "thread", shard_channel=channel,
                                 pipeline=args.pipeline_depth > 0, pipeline_depth=max(1, args.pipeline_depth),
                                 backend_priority=args.backend_priority, encoder_priority=args.encoder_priority,
                                 backend_stages=args.backend_stages, solve_priority=args.solve_priority)
        # the stream: RGB frames rendered on the device, resident in HBM before the clock starts
        shp = torch.tensor([[H, W]])
        self.frames = []
        for lo in range(0, total_frames, 16):
            k = self.base + args.stride * torch.arange(lo, min(lo + 16, total_frames), device=dev)
            img = self.model.room.rgb(k)
            for j in range(img.shape[0]):
                self.frames.append(Frame(lo + j, img[j:j + 1].clone(), shp, shp, None))
        self.host_imgs = None
        if args.host_frames:
            self.host_imgs = [f.img.cpu().pin_memory() for f in self.frames]
            for f in self.frames:
                f.img = None
        self.pos = 0

    def run(self, n):
        if self.host_imgs is not None:      # H2D inside the timed region, in front of the frames' first use
            for k in range(self.pos, min(self.pos + n, len(self.frames))):
                self.frames[k].img = self.host_imgs[k].to(self.dev, non_blocking=True)
                self.host_imgs[k] = None
        if self.args.tracking_priority != 0:
            if not hasattr(self, "_trk_stream"):
                self._trk_stream = torch.cuda.Stream(device=self.dev, priority=self.args.tracking_priority)
                self._trk_stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self._trk_stream):
                self.system.run(self.frames, self.pos, self.pos + n, release=True)
        else:
            self.system.run(self.frames, self.pos, self.pos + n, release=True)
        self.pos += n

    def drain(self):
        self.system.drain()

    def graph(self):
        fg = self.system.factor_graph
        return len(self.system.keyframes), int(fg.ii.numel())


def dominant_shape(B):
    """The single kernel with the largest share of the step (profiles/r02_bench_kernel_stats.csv): the bf16 GEMM of the
    encoder's fc1 (768 B x 4096 x 1024, GELU epilogue) at the frame group's row count."""
    return 768 * B, 4096, 1024


def isolated_dominant_us(L, m, M, N, K, dev, iters=50):
    """The dominant kernel alone on the GPU (nothing else running), event-timed: what the live figure would be without
    the other streams' kernels sharing the chip (the profiles under profiles/ are taken this way)."""
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Wt = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
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
    return 1e3 * e0.elapsed_time(e1) / iters


def hbm_kernel_probes(ses, dev):
    """The HBM-bound kernels of the step, each alone on the GPU on one full-resolution room pair, event-timed: algorithmic
    bytes (SURVEY §8d / DESIGN.md per-unit figures) / launch time against the 8 TB/s HBM peak.  Isolated, after the clock
    has stopped (inside the loop their launches are a few per frame and share the chip with the network)."""
    import mast3r_slam_backends as be
    from mast3r_slam import matching
    from mast3r_slam.config import config

    room = ses.model.room
    ki, kj = torch.tensor([3.0], device=dev), torch.tensor([0.0], device=dev)
    a, b = room.pair_fused(ki, kj)
    mc = config["matching"]
    rays, pts, p0 = matching.prep_for_iter_proj(a["pts3d"], b["pts3d"], None)
    D11 = a["desc"].half().contiguous()
    D21 = b["desc"].reshape(1, H * W, -1).half().contiguous()
    p_new, conv = be.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"])
    p1 = p_new.long().contiguous()

    def timed(fn, n=20):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    out = []
    def add(name, nbytes, us):
        out.append({"kernel": name, "algorithmic_bytes": nbytes, "us_per_launch_isolated": us,
                    "achieved_GBps": nbytes / us * 1e-3, "frac_of_8TBps": nbytes / us * 1e-3 / PEAK_HBM_GBS})
    add("prep_iter_proj_kernel (rays + gradients, 1 pair direction)", H * W * (24 + 36 + 12 + 8),
        timed(lambda: matching.prep_for_iter_proj(a["pts3d"], b["pts3d"], None)))
    add("iter_proj_kernel (1 pair direction)", 12.8e6,
        timed(lambda: be.iter_proj(rays, pts, p0, mc["max_iter"], mc["lambda_init"], mc["convergence_thresh"])))
    add("refine_matches_kernel<24> (1 pair direction; bound by L2 gathers + its sequential IEEE-half add chain, not HBM: profiles/r02_pmc_refine_matches.json)", 25.2e6,
        timed(lambda: be.refine_matches(D11, D21, p1, mc["radius"], mc["dilation_max"])))
    # GN edge kernels on a small graph at full resolution
    from mast3r_slam import synthetic

    g = synthetic.make_graph(n_kf=3, h=H, w=W, seed=3, stride=3, extra_edges=1)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in g.items() if isinstance(v, np.ndarray)}
    E = int(d["ii"].numel())
    valid_pts = float(d["valid_match"].float().sum())
    lc = config["local_opt"]
    blk = lambda: be.gn_blocks("rays", d["Twc"], d["Xs"], d["Cs"], None, d["ii"], d["jj"], d["idx_ii2jj"], d["valid_match"],
                               d["Q"], lc["sigma_ray"], lc["sigma_dist"], lc["C_conf"], lc["Q_conf"])
    t_once = timed(blk, 10)       # begin + compact + one accumulate + reduce
    full = lambda it: be.gauss_newton_rays(d["Twc"].clone(), d["Xs"], d["Cs"], d["ii"], d["jj"], d["idx_ii2jj"],
                                           d["valid_match"], d["Q"], lc["sigma_ray"], lc["sigma_dist"], lc["C_conf"],
                                           lc["Q_conf"], it, lc["delta_norm"])
    t1, t9 = timed(lambda: full(1), 5), timed(lambda: full(9), 5)
    per_it = (t9 - t1) / 8.0       # accumulate + solve of one iteration of this tiny graph
    add(f"gn_compact + gn_accum + reduce, first pass ({E} directed edges: launch-latency bound at this size)",
        E * H * W * 45.0 + valid_pts * 32.0, t_once)
    add(f"gn_accum_kernel per iteration incl. the 14-unknown solve ({E} directed edges, {valid_pts / (E * H * W):.2f} of the points "
        "survive; latency bound at this size - at 980 edges the kernel streams 4.1 TB/s, profiles/r02_gn_125kf_kernel_stats.csv)",
        valid_pts * 28.0, per_it)
    return out


def cpu_baseline(args, graph_kfs, graph_edges, edges_per_kf, kf_every):
    """The oracle (kind = "port") timed on the host cores for a BOUNDED sample of the same workload:
    one tracked frame (torch-CPU fp32 network restatement + C matching + numpy tracking GN) plus one
    keyframe's backend with ONE symmetric edge direction of network (scaled to the measured edges per keyframe), one GN
    iteration of the C restatement on a 4-keyframe graph (scaled to the measured graph) and the C TSDF on 40 000 points."""
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
    t_gn_edge_iter = (time.perf_counter() - t0) / e_small          # per directed edge per iteration (OpenMP threads)
    t_gn = t_gn_edge_iter * 2 * graph_edges * lc["max_iters"]        # + the dense LL^T, negligible beside the edge pass
    note(f"GN {t_gn_edge_iter:.3f}s per edge-iteration; TSDF")
    Xw = synthetic.sim3_act(Tk, Xk.astype(np.float64))
    sel = np.random.default_rng(0).permutation(H * W)[:40000]
    vol = oracle.TSDFVolume(0.03, 0.12)
    t0 = time.perf_counter()
    vol.integrate(Xw[sel].astype(np.float32), np.full(40000, 2.0), Tk[:3].astype(np.float32))
    t_tsdf = time.perf_counter() - t0
    per_frame = t_net_frame + t_match + t_track
    per_kf = edges_per_kf * t_net_edge + 2 * edges_per_kf * t_match + t_gn + t_tsdf
    sec_per_frame = per_frame + per_kf / kf_every
    return dict(value=1.0 / sec_per_frame, unit="frames/s", cores=cores, kind="port",
                sample=("1 tracked frame (torch-CPU fp32 network %.1fs, C matching %.2fs, numpy tracking GN %.2fs) + "
                        "1 keyframe backend extrapolated from 1 decoder+heads pass to %.1f edges, 1 GN iteration on %d "
                        "directed edges scaled to the measured %d-keyframe / %d-directed-edge graph x %d iterations, "
                        "40k-point TSDF %.2fs; one keyframe per %.1f frames as measured"
                        % (t_net_frame, t_match, t_track, edges_per_kf, e_small, graph_kfs, 2 * graph_edges,
                           lc["max_iters"], t_tsdf, kf_every)))


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, one process per
    GPU, rendezvous on 127.0.0.1) BEFORE this process has made any GPU call, hand their output through and return their
    exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def tsdf_cfgs(args):
    from mast3r_slam.config import config

    if args.no_tsdf:
        return None, None
    # pre_icp_iters / max_iterations 0: the reference's TSDF pose refinement steps along the UNIT gradient with the
    # truncation-normalised residual (tsdf_optimizer.py:94-124), an 8x overshoot at trunc_dist 0.12 that throws the
    # keyframe poses off and sends tracking into relocalisation on this scene (tools/slam_room_probe.py); its
    # kernels are parity-tested on their own (tests/test_tsdf_gpu.py), the loop runs fusion + re-fusion only
    tg = dict(config["tsdf_global"], enabled=True, hash_capacity=1 << 22, pre_icp_iters=0, max_iterations=0)
    tr = dict(config["tsdf_refine"], enabled=True)
    return tg, tr


def measure_sharded_backend(args, rank, world, dev, ranks_seen):
    """ONE session, backend sharded over the `world` ranks (mast3r_slam/shard.py).  Rank 0 runs the product loop exactly
    as in the replica measurement (same frames, same schedule), the other ranks serve.  Returns the report (rank 0)."""
    from mast3r_slam.shard import OP_PAUSE, OP_STOP, BackendShard, ShardChannel

    ch = ShardChannel(dev)
    preroll = args.preroll if args.preroll >= 0 else (500 if args.steps < 500 else 0)
    total = preroll + args.warmup + args.steps
    phases = [p for p in (preroll, args.warmup) if p] + [args.steps]
    if rank == 0:
        ses = Session(args, 0, world, dev, total, channel=ch)
        marks = []
        for n in phases:
            barrier(world)
            t0 = time.perf_counter()
            ses.run(n)
            ses.drain()
            with ch.lock:
                ch.announce(OP_PAUSE)
            barrier(world)
            marks.append((time.perf_counter() - t0, ses.graph(), dict(ses.system.stats)))
        ses.system.shutdown()             # still announces (the voxel tables report dropped samples): before the stop
        with ch.lock:
            ch.announce(OP_STOP)
        elapsed = marks[-1][0]
    else:
        from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, random_state_dict
        from mast3r_slam.synthetic_gpu import RoomGeometryModel

        net = None
        if not args.no_network:
            mc = Mast3rConfig(enc_depth=max(1, round(24 * args.depth_scale)), dec_depth=12)
            sd = random_state_dict(mc, seed=0)
            net = Mast3rHIP(sd, mc, device=dev, use_graphs=args.graphs)
            del sd
        model = RoomGeometryModel(net, dev, H, W, n_frames=1000, seed=0)
        shard = BackendShard(model, dev, ch, tsdf_global_cfg=tsdf_cfgs(args)[0])
        elapsed = 0.0
        for n in phases:
            barrier(world)
            t0 = time.perf_counter()
            assert shard.serve() == "pause"
            barrier(world)
            elapsed = time.perf_counter() - t0
        assert shard.serve() == "stop"
    import torch.distributed as dist

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    if rank != 0:
        return None
    (kf0, e0), st0 = (marks[-2][1], marks[-2][2]) if len(marks) > 1 else ((0, 0), {})
    (kf1, e1), st1 = marks[-1][1], marks[-1][2]
    from mast3r_slam.config import config

    return {"value": args.steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / args.steps, "scaling": "strong",
            "ranks": world, "ranks_reported_by_collective_library": ranks_seen,
            "what": "ONE session: rank 0 tracks + drives; keyframe-pair inference + matching split over the ranks "
                    "(all-gather of the matches), global GN edges split over the ranks with one all-reduce(sum) of the "
                    "normal-equation blocks per iteration, global TSDF voxels on their owner ranks (replicated point list)",
            "keyframes": [kf0, kf1], "undirected_edges": [e0, e1],
            "gn_iterations_per_solve": int(config["local_opt"]["max_iters"]),
            "allreduce_bytes_per_gn_iteration": (4 * 49 + 2 * 7) * 4 * 2 * e1,
            "broadcast_bytes_total": ch.bytes_broadcast,
            "announcements": {str(k): v for k, v in sorted(ch.announced.items())},
            "relocalised": st1.get("relocalised", 0) - st0.get("relocalised", 0)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # started without a launcher: be the launcher
        sys.exit(spawn_ranks(args))
    if os.environ.get("BENCH_WATCHDOG"):   # debug: dump every thread's stack and exit if the run takes longer than this
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG"]), exit=True)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU "
              "(python bench.py --gpus N starts them itself)", file=sys.stderr)
        sys.exit(2)
    rank, world, dev = dist_setup(args)
    import mslam_hip

    L = mslam_hip.lib()
    mslam_hip.check(L.mslam_device_check(), "device_check")
    ranks_seen = 1
    if world > 1:   # the rank count the collective library itself reports (RCCL over xGMI when launched one rank per GPU)
        import torch.distributed as dist

        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    mode = args.mode
    if world == 1 and mode == "shard-backend":
        print("error: --mode shard-backend needs --gpus > 1", file=sys.stderr)
        sys.exit(2)
    out = None
    if mode in ("auto", "replicas"):
        out = measure_replicas(args, rank, world, dev, L, mslam_hip, ranks_seen)
    if world > 1 and mode in ("auto", "shard-backend"):
        if out is not None:
            torch.cuda.empty_cache()
        sb = guarded_sharded_backend(args, rank, world, dev, ranks_seen, out, mode == "auto")
        if rank == 0:
            if out is None:   # --mode shard-backend: the sharded session is the line's value
                out = {"metric": "SLAM frames/sec (infer+match+TSDF+GN) @512x384", "value": sb["value"], "unit": "frames/s",
                       "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sb["ms_per_step"],
                       "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
                       "data": "synthetic",
                       "config": {"workload": "ONE synthetic 512x384 session through the product loop, backend sharded over "
                                              f"{world} ranks (see sharded_backend)", "frame_group": args.frame_group}}
            out["sharded_backend"] = sb
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def guarded_sharded_backend(args, rank, world, dev, ranks_seen, out, replicas_done):
    """measure_sharded_backend behind a watchdog.  The replica figure (`out`, the line's `value`) is already measured when
    the sharded session starts; a session that raises on one rank leaves the others inside a collective, and one that hangs
    (this path has only ever run on one card) would take the whole line with it.  Every rank therefore arms the same timer
    (BENCH_SHARD_TIMEOUT seconds, default 300) at the same barrier: if the session has not finished by then - or raises -
    rank 0 prints the line with `sharded_backend: {"error": ...}` and every rank leaves with exit code 0 (3 when the
    sharded session WAS the requested measurement, --mode shard-backend)."""
    import threading
    import traceback

    timeout = float(os.environ.get("BENCH_SHARD_TIMEOUT", "300"))

    def bail(reason):
        if rank == 0:
            print(f"[bench] sharded session abandoned: {reason}", file=sys.stderr, flush=True)
            if out is not None:
                out["sharded_backend"] = {"error": reason}
                print(json.dumps(out), flush=True)
        os._exit(0 if replicas_done else 3)     # (`out` exists on rank 0 only)

    barrier(world)
    timer = threading.Timer(timeout, bail, args=(f"did not finish within {timeout:.0f} s",))
    timer.daemon = True
    timer.start()
    try:
        sb = measure_sharded_backend(args, rank, world, dev, ranks_seen)
    except Exception as e:      # this rank leaves; the others run into their own timers
        traceback.print_exc()
        timer.cancel()
        bail(f"rank {rank}: {type(e).__name__}: {e}")
    timer.cancel()
    return sb


def measure_replicas(args, rank, world, dev, L, mslam_hip, ranks_seen):
    preroll = args.preroll if args.preroll >= 0 else (500 if args.steps < 500 else 0)
    total = preroll + args.warmup + args.steps
    t_setup = time.perf_counter()
    ses = Session(args, rank, world, dev, total)
    B = ses.system.frame_group
    # construction-time priming + pre-roll (not warm-up steps of the contract): arenas, streams and kernel attributes
    # exist, and the keyframe graph has the size the timed steps are quoted on
    if preroll:
        ses.run(preroll)
        ses.drain()
    barrier(world)
    kf_pre, e_pre = ses.graph()
    if rank == 0:
        print(f"[bench] setup + pre-roll of {preroll} frames: {time.perf_counter() - t_setup:.1f}s -> {kf_pre} keyframes, "
              f"{e_pre} undirected edges", file=sys.stderr)
    ses.run(args.warmup)
    ses.drain()
    barrier(world)
    kf0, e0 = ses.graph()
    st0 = dict(ses.system.stats)
    st0["verdict_wait_s"] = ses.system.tracker.verdict_wait_s
    st0["backend_phase_s"] = dict(st0.get("backend_phase_s", {}))
    rows0 = (ses.model.enc_rows, ses.model.dec_rows)
    # the look-ahead encoder's batch: calls do not cross the end of a run() segment, so a timed region shorter than the
    # batch sees ONE call of `steps` frames
    M, N, K = dominant_shape(min(max(B, ses.system.encoder_group), max(1, args.steps)))
    mslam_hip.check(L.mslam_gemm_profile_begin(M, N, K, 8192), "gemm_profile_begin")
    prof = None
    if os.environ.get("BENCH_CPROFILE"):   # debug: where the frontend thread's host time goes
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    ses.run(args.steps)
    t_enqueued = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)   # host time of the frontend loop (it reads one tracking verdict per frame)
    ses.drain()                             # every queued keyframe task has been issued ...
    barrier(world)                          # ... and (device-wide synchronise inside) has finished
    elapsed = time.perf_counter() - t0
    import ctypes

    avg_us, min_us, nsamp = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    mslam_hip.check(L.mslam_gemm_profile_end(ctypes.byref(avg_us), ctypes.byref(min_us), ctypes.byref(nsamp)),
                    "gemm_profile_end")
    kf1, e1 = ses.graph()
    st1 = dict(ses.system.stats)
    st1["verdict_wait_s"] = ses.system.tracker.verdict_wait_s
    iso_us = isolated_dominant_us(L, mslam_hip, M, N, K, dev) if not args.no_network else 0.0
    hbm = hbm_kernel_probes(ses, dev)
    enc_rows, dec_rows = ses.model.enc_rows - rows0[0], ses.model.dec_rows - rows0[1]
    if world > 1:
        import torch.distributed as dist

        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        t_wait = st1.get("verdict_wait_s", 0.0) - st0.get("verdict_wait_s", 0.0)
        print(f"[bench] frontend loop host time {1e3 * t_enqueued / args.steps:.2f} ms/step of {1e3 * elapsed / args.steps:.2f} ms/step "
              f"= {1e3 * (t_enqueued - t_wait) / args.steps:.2f} ms/step of enqueue work + {1e3 * t_wait / args.steps:.2f} ms/step waiting for "
              "the device (tracking verdicts)", file=sys.stderr)
        fps = args.steps * world / elapsed
        new_kf, new_e = kf1 - kf0, e1 - e0
        if st1.get("backend_phase_s"):   # MSLAM_BACKEND_PROFILE=1: the keyframe task's chain, phase by phase (whole session)
            ph, ph0 = st1["backend_phase_s"], st0.get("backend_phase_s", {})
            print(f"[bench] backend phases, ms per keyframe task in the timed region ({new_kf} tasks, {kf0}->{kf1} keyframes): " +
                  ", ".join(f"{k} {1e3 * (v - ph0.get(k, 0.0)) / max(1, new_kf):.2f}" for k, v in ph.items()), file=sys.stderr)
        # network FLOP actually launched inside the timed region (rows of every encoder / decoder+heads call)
        gflop = 0.0 if args.no_network else enc_rows * ses.gf_enc + dec_rows * (GF_DEC + 2 * GF_HEAD)
        dom_gflop = 2e-9 * M * N * K
        dom_tflops = dom_gflop * 1e3 / avg_us.value if nsamp.value else 0.0
        traffic = None
        import glob

        # PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE of the same launch, separate passes (tools/pmc_probe.sh): the committed
        # record of THIS shape, newest round first
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dominant_kernel*.json")), reverse=True):
            try:
                tj = json.load(open(tpath))
                if tj.get("shape") == [M, N, K] and tj.get("hbm_bytes_per_launch"):
                    traffic = tj.get("hbm_bytes_per_launch")
                    break
            except Exception:
                pass
        out = {
            "metric": "SLAM frames/sec (infer+match+TSDF+GN) @512x384", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "synthetic 512x384 RGB-D stream through the product loop (SlamSystem.run): tracked frame "
                                   f"every step, real keyframe decisions ({new_kf} new keyframes in the timed {args.steps} steps), "
                                   f"backend per keyframe over the whole graph: {kf0}->{kf1} keyframes, {e0}->{e1} undirected "
                                   f"edges ({2 * e1} directed) incl. retrieval k={args.retrieval_k} "
                                   f"({'pose-proximity stand-in' if args.retriever == 'pose' else 'ASMK RetrievalDatabase, random head + codebook'}), 40k-point global TSDF fuse + "
                                   "budgeted re-fusion (TSDF pose refinement off: the reference's step overshoots 8x and "
                                   "breaks tracking on this scene), local TSDF block refinement"
                                   + ("; frames uploaded from pinned host memory inside the timed region (PCIe-inclusive)" if args.host_frames else "")
                                   + (f"; {preroll} frames pre-rolled untimed so that the graph has the mean size of the "
                                      "1 000-frame schedule" if preroll else "; no pre-roll: the graph grows from the first frame"),
                       "weights": "random-init ViT-L/12+12 MASt3R architecture (no checkpoint offline); geometry from the "
                                  "procedural room stand-in, rendered on the device inside the timed region",
                       "backend": "inline" if args.no_backend_thread else (
                           "own host thread + stream (as the reference's backend process)" if args.backend_stages < 2 else
                           f"{args.backend_stages} host threads + streams: graph stage (retrieval, pair inference, matching) of keyframe k+1 "
                           "beside the solve stage of keyframe k" + (" beside the TSDF fusion + local refinement of keyframe k-1" if args.backend_stages >= 3 else "") +
                           " (the reference's backend is a process of its own)"),
                       "frame_group": B, "encoder_group": ses.system.encoder_group, "camera_path_stride": args.stride, "match_frac_thresh": args.kf_thresh,
                       "stats": {"keyframes": kf1, "new_keyframes": new_kf, "new_edges": new_e,
                                 "decoded_rows_tracking": st1["decoded_rows"] - st0["decoded_rows"],
                                 "void_rows": st1["void_rows"] - st0["void_rows"],
                                 "encoder_rows": enc_rows, "decoder_rows_total": dec_rows,
                                 "refine_blocks": st1.get("refine_blocks", 0) - st0.get("refine_blocks", 0),
                                 "relocalised": st1["relocalised"] - st0["relocalised"],
                                 "replayed_frames": st1.get("replayed_frames", 0) - st0.get("replayed_frames", 0),
                                 "frontend_host_ms_per_step": 1e3 * t_enqueued / args.steps,      # whole loop iteration on the host
                                 "frontend_enqueue_ms_per_step": 1e3 * (t_enqueued - t_wait) / args.steps,
                                 "frontend_wait_for_device_ms_per_step": 1e3 * t_wait / args.steps},
                       "parallelism": f"{world} independent session(s), one per GPU (replicas; {ranks_seen} rank(s) reported by the collective library)"},
            "roofline": {"bound": "mfma", "achieved": dom_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom_tflops / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "kernel": f"gemm_bf16_kernel, encoder fc1 {M}x{N}x{K} + GELU (largest share of the step)",
                         "gflop_per_launch": dom_gflop, "us_per_launch_avg": avg_us.value, "us_per_launch_min": min_us.value,
                         "launches_timed": nsamp.value,
                         "us_per_launch_isolated": iso_us,
                         "tflops_isolated": dom_gflop * 1e3 / iso_us if iso_us else 0.0,
                         "method": "HIP events around every launch of this shape inside the timed region, on its launch stream "
                                   "(mslam_gemm_profile_begin/end)",
                         "network_tflops_over_timed_region": gflop / (1e3 * elapsed) if elapsed > 0 else 0.0,
                         "network_gflop_per_step": gflop / args.steps,
                         "hbm_bound_kernels": hbm},
        }
        if not args.no_cpu_baseline and world == 1:   # the host baseline is reported on rank 0 at N = 1 only
            kfs_mean = max(2, (kf0 + kf1) // 2)
            edges_mean = max(1, (e0 + e1) // 2)
            out["cpu_baseline"] = cpu_baseline(args, kfs_mean, edges_mean, max(1.0, new_e / max(1, new_kf)),
                                               args.steps / max(1, new_kf))
    ses.system.shutdown()
    del ses
    return out if rank == 0 else None




if __name__ == "__main__":
    main()
