"""ONE session's backend over the N GPUs of a node (north_star: "keyframe-pair inference batches and TSDF voxel blocks
shard across the 8 GPUs of one node with RCCL all-reduce over xGMI for the global_opt.py JtJ accumulate"; BASELINE
configs 4 and 5).  The reference runs the backend as one process on one device (main.py:73-163): it loops over the new
keyframe pairs serially (mast3r_utils.py:83-115), solves the whole graph on that device (global_opt.py:123-164) and
keeps the TSDF in one python dict (tsdf/global_manager.py:177-226).

Roles.  Rank 0 is the DRIVER: it runs the whole product loop (`SlamSystem`: tracking, retrieval, keyframe decisions,
local TSDF) and owns the keyframe store.  Ranks 1..N-1 are SHARDS (`BackendShard.serve()`): they hold what the sharded
stages need and nothing else - the encoder tokens of every keyframe (immutable), the GN-ready pointmaps /
confidences of every keyframe (re-sent when they change), the factor-graph edge lists (grown by the same deterministic
calls on every rank), one shard of the voxel table.

The collective layer underneath is SPMD - `global_opt.match_symmetric_sharded`, `global_opt.gauss_newton_sharded`, the
sharded `TSDFVolume` / `TSDFPoseOptimizer` methods are called by every rank with identical arguments
(tests/shard_worker.py drives them that way).  `ShardChannel` is what turns the driver's ordinary method calls into those
SPMD calls: before the driver enters a sharded method it ANNOUNCES it - one small broadcast with the op code and the
scalar arguments, followed by broadcasts of the tensors the shards do not have yet - and the shards, blocked in
`serve()`, make the same call.  Per keyframe that is: the new keyframe's tokens (3.1 MB) once, the pointmaps of the
keyframes that changed since the last solve (3.1 MB each: the new one, the previous one, whatever the local refiner
touched), the poses (32 B per keyframe), the 40 000 fused points (1.3 MB) - against 2 x 437 GFLOP of pair inference per
edge and one all-reduce of the normal-equation blocks (840 B per directed edge) per GN iteration.

Every collective of a session is issued by ONE host thread at a time on the driver (the backend thread; the tracking
thread only during relocalisation, when the backend is drained), so the ranks always agree on the order."""
import struct
import threading

import torch

OP_STOP, OP_PAUSE, OP_ADD_FACTORS, OP_POINTMAPS, OP_SOLVE, OP_TSDF_FUSE, OP_TSDF_MAINTAIN, OP_TSDF_REFINE, OP_TSDF_NEQ, \
    OP_TSDF_QUERY, OP_TSDF_VOXELS = range(11)
POINTMAPS_PER_ANNOUNCEMENT = 32
HDR = 256          # int64 words per announcement


def _f2i(x):
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _i2f(i):
    return struct.unpack("<d", struct.pack("<q", int(i)))[0]


class ShardChannel:
    """Announcements driver -> shards over a torch.distributed group (RCCL when the ranks sit on their own cards, gloo when
    they share one: the rehearsal mode).  `announce(op, ints, floats)` on the driver pairs with `receive()` on a shard."""

    def __init__(self, device, group=None):
        import torch.distributed as dist

        self.dist, self.group, self.device = dist, group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.is_driver = self.rank == 0
        self.lock = threading.RLock()
        self.announced = {}         # op -> count (driver), for the bench's report
        self.bytes_broadcast = 0

    def announce(self, op, ints=(), floats=()):
        assert self.is_driver and 3 + len(ints) + len(floats) <= HDR, (op, len(ints), len(floats))
        words = [int(op), len(ints), len(floats)] + [int(i) for i in ints] + [_f2i(f) for f in floats]
        words += [0] * (HDR - len(words))
        hdr = torch.tensor(words, dtype=torch.int64).to(self.device)
        self.dist.broadcast(hdr, src=0, group=self.group)
        self.announced[op] = self.announced.get(op, 0) + 1

    def receive(self):
        hdr = torch.empty(HDR, dtype=torch.int64, device=self.device)
        self.dist.broadcast(hdr, src=0, group=self.group)
        w = hdr.cpu().tolist()
        ni, nf = w[1], w[2]
        return w[0], w[3:3 + ni], [_i2f(i) for i in w[3 + ni:3 + ni + nf]]

    def bcast(self, t):
        """In place on every rank; returns t.  (Under inference mode: the driver's keyframe tensors were produced under
        it, and a collective counts as an in-place update of its argument.)"""
        with torch.inference_mode():
            self.dist.broadcast(t, src=0, group=self.group)
        self.bytes_broadcast += t.numel() * t.element_size()
        return t

    def all_reduce(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class _TokenStore:
    """What FactorGraph.add_factors reads from a keyframe (global_opt.py:33-41): feat, pos, img_true_shape."""

    class _Kf:
        __slots__ = ("feat", "pos", "img_true_shape", "frame_id")

    def __init__(self):
        self.kfs = {}

    def __getitem__(self, idx):
        return self.kfs[int(idx)]

    def __len__(self):
        return (max(self.kfs) + 1) if self.kfs else 0


class BackendShard:
    """A shard rank of one session: blocks in serve() and mirrors the driver's sharded calls."""

    def __init__(self, model, device, channel, tsdf_global_cfg=None, use_calib=False):
        from mast3r_slam.global_opt import FactorGraph

        self.model, self.device, self.ch = model, torch.device(device), channel
        self.tokens = _TokenStore()
        self.factor_graph = FactorGraph(model, self.tokens, None, device, shard_edges=True, channel=channel)
        self.Xs, self.Cs = {}, {}          # keyframe id -> GN-ready pointmap (HW,3) / average confidence (HW,1)
        self.volume = self.optimizer = None
        if tsdf_global_cfg is not None and tsdf_global_cfg.get("enabled", False):
            from mast3r_slam.tsdf import TSDFVolume
            from mast3r_slam.tsdf.tsdf_optimizer import TSDFPoseOptimizer

            cfg = tsdf_global_cfg
            self.volume = TSDFVolume(voxel_size=cfg.get("voxel_size", 0.03), truncation=cfg.get("trunc_dist", 0.12),
                                     max_weight=cfg.get("max_weight", 100.0), min_weight=cfg.get("min_tsdf_weight", 1.0e-3),
                                     capacity=int(cfg.get("hash_capacity", 1 << 22)), device=device,
                                     shard_id=channel.rank, num_shards=channel.world, channel=channel)
            self.optimizer = TSDFPoseOptimizer(self.volume, None, cfg, use_calib, device)
        self.served = {}

    def serve(self):
        """Mirror announcements until the driver pauses (returns "pause": the caller may synchronise / take the time and
        call serve() again) or stops (returns "stop")."""
        ch = self.ch
        while True:
            op, ints, floats = ch.receive()
            self.served[op] = self.served.get(op, 0) + 1
            if op == OP_STOP:
                return "stop"
            if op == OP_PAUSE:
                return "pause"
            if op == OP_ADD_FACTORS:
                self._add_factors(ints, floats)
            elif op == OP_POINTMAPS:
                hw, n = ints[:2]
                pk = ch.bcast(torch.empty((n, hw, 4), dtype=torch.float32, device=self.device))
                for r, k in enumerate(ints[2:2 + n]):
                    self.Xs[k], self.Cs[k] = pk[r, :, :3].contiguous(), pk[r, :, 3:4].contiguous()
            elif op == OP_SOLVE:
                self._solve(ints, floats)
            elif op == OP_TSDF_FUSE:
                n, = ints
                pk = ch.bcast(torch.empty((n + 1, 4), dtype=torch.float64, device=self.device))
                self.volume.integrate(pk[:n, :3].float(), pk[:n, 3].contiguous(), pk[n, :3].float(), return_fused=False)
            elif op == OP_TSDF_MAINTAIN:
                self.volume.maintain(reserve=ints[0])
            elif op == OP_TSDF_REFINE:
                n, iters = ints
                pk = ch.bcast(torch.empty((n + 2, 4), dtype=torch.float32, device=self.device))
                from lietorch_hip import Sim3

                pose = Sim3(torch.cat((pk[n], pk[n + 1])).reshape(1, 8).clone())
                self.optimizer.refine_pose(pose, pk[:n, :3].contiguous(), pk[:n, 3].contiguous(), iterations=iters)
            elif op == OP_TSDF_NEQ:
                n, = ints
                pk = ch.bcast(torch.empty((n, 4), dtype=torch.float32, device=self.device))
                self.optimizer.normal_equations(pk[:, :3].contiguous(), pk[:, 3].contiguous())
            elif op == OP_TSDF_QUERY:
                n, = ints
                pts = ch.bcast(torch.empty((n, 3), dtype=torch.float32, device=self.device))
                self.volume.query_batch(pts)
            elif op == OP_TSDF_VOXELS:
                self.volume.voxels()
            else:
                raise RuntimeError(f"BackendShard: unknown announcement {op}")

    # ------------------------------------------------------------------ the two factor-graph calls
    def _add_factors(self, ints, floats):
        n, is_reloc, h, w, ntok, n_new = ints[:6]
        ii, jj = ints[6:6 + n], ints[6 + n:6 + 2 * n]
        new = ints[6 + 2 * n:6 + 2 * n + 2 * n_new]
        for k in range(n_new):
            kf = _TokenStore._Kf()
            kf.frame_id = new[2 * k + 1]
            kf.feat = self.ch.bcast(torch.empty((1, ntok, 1024), dtype=torch.float32, device=self.device))
            kf.pos = self.ch.bcast(torch.empty((1, ntok, 2), dtype=torch.int64, device=self.device))
            kf.img_true_shape = torch.tensor([[h, w]])      # host, as the driver holds it (read with .tolist())
            self.tokens.kfs[new[2 * k]] = kf
        self.factor_graph.add_factors(ii, jj, floats[0], is_reloc=bool(is_reloc))

    def _solve(self, ints, floats):
        kind_id, P, height, width = ints[:4]
        fg = self.factor_graph
        ids = fg.get_unique_kf_idx().tolist()          # the same edge lists on every rank -> the same keyframe rows
        assert len(ids) == P, (len(ids), P)
        pose_data = self.ch.bcast(torch.empty((P, 8), dtype=torch.float32, device=self.device))
        kind = ("rays", "calib", "points")[kind_id]
        K = torch.tensor(floats[:9], dtype=torch.float32, device=self.device).reshape(3, 3) if kind == "calib" else None
        ii, jj, sources = fg.two_way_sources()
        job = dict(kind=kind, pin=fg.cfg["pin"], K=K, height=height, width=width, pose_data=pose_data,
                   Xs=torch.stack([self.Xs[k] for k in ids]), Cs=torch.stack([self.Cs[k] for k in ids]),
                   edges=(ii, jj, sources))
        fg.run_solve(job)
