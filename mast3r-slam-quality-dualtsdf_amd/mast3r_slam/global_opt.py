"""Mirror of mast3r_slam/global_opt.py (FactorGraph, lines 12-223): edge bookkeeping, symmetric
matching per edge batch, two-way edge preparation and the calls into the native Gauss-Newton.

Multi-GPU (new; the reference is single-device): when torch.distributed is initialised and
`shard_edges=True`, the ranks hold the same keyframes (replicated; `broadcast_keyframe` ships a new keyframe's
encoder tokens, 3.1 MB, from the rank that tracked it) and share the backend of ONE session:
* add_factors: each rank decodes + matches ITS slice of the new keyframe pairs (the reference loops over them
  serially, mast3r_utils.py:83-115); the match results (idx, valid, Q: ~2.5 MB per pair direction) are
  all-gathered, so every rank appends the same edges;
* solve: each rank accumulates the normal-equation blocks of ITS slice of the directed edges, the reference-layout
  block buffers Hs[4,E,7,7] / gs[2,E,7] are summed with ONE all-reduce per GN iteration (RCCL over xGMI on MI355X;
  other ranks' slots are zero so the sum is exact and every rank gets bit-identical blocks), and every rank runs the
  same fp64 solve + retraction - no broadcast."""
import threading

import torch

import mast3r_slam_backends
import mslam_hip as _m
from lietorch_hip import Sim3
from mast3r_slam.config import config
from mast3r_slam.geometry import constrain_points_to_ray
from mast3r_slam.mast3r_utils import mast3r_match_symmetric


class FactorGraph:
    def __init__(self, model, frames, K=None, device="cuda", shard_edges=False, channel=None):
        self.model = model
        self.frames = frames
        self.device = device
        self.cfg = config["local_opt"]
        # edge lists: row-appendable buffers (capacity doubling) instead of the reference's torch.cat per keyframe
        # (global_opt.py:92-99), which copies every [E, HW] array - 0.7 GB each at 440 edges - whenever edges are added
        # and again for every solve; the attributes below are views of the filled part
        self._rows = {k: _Rows(dt, self.device) for k, dt in (
            ("ii", torch.long), ("jj", torch.long), ("idx_ii2jj", torch.long), ("idx_jj2ii", torch.long),
            ("valid_match_j", torch.bool), ("valid_match_i", torch.bool), ("Q_ii2jj", torch.float32),
            ("Q_jj2ii", torch.float32))}
        self.window_size = self.cfg["window_size"]
        self.K = K
        self.last_unique_kf_idx = None
        self.shard_edges = shard_edges
        self.reuse_tracking_decode = True
        self.reused_rows = 0
        # driver / shard roles of one session (mast3r_slam/shard.py): the driver announces its sharded calls so that the
        # shard ranks make them too; None = every rank calls the sharded methods itself (SPMD)
        self.channel = channel
        self.group = channel.group if channel is not None else None
        self._sent_tokens = {}     # keyframe index -> frame_id whose encoder tokens the shards hold
        self._sent_stamp = {}      # keyframe index -> store stamp of the pointmap the shards hold
        # a two-stage backend appends the next keyframe's edges (graph stage) while the solve stage reads the lists
        self.lock = threading.Lock()
        self._pm_cache = None      # stacked pointmaps / confidences of the keyframes, refreshed row by row (prepare_solve)

    # ------------------------------------------------------------------
    def add_factors(self, ii, jj, min_match_frac, is_reloc=False):
        """global_opt.py:32-101."""
        kf_ii = [self.frames[idx] for idx in ii]
        kf_jj = [self.frames[idx] for idx in jj]
        feat_i = torch.cat([kf.feat for kf in kf_ii])
        feat_j = torch.cat([kf.feat for kf in kf_jj])
        pos_i = torch.cat([kf.pos for kf in kf_ii])
        pos_j = torch.cat([kf.pos for kf in kf_jj])
        shape_i = [kf.img_true_shape for kf in kf_ii]
        shape_j = [kf.img_true_shape for kf in kf_jj]
        if self._driver:
            self._announce_add_factors(ii, jj, min_match_frac, is_reloc)
        if self.shard_edges and torch.distributed.is_available() and torch.distributed.is_initialized():
            res = match_symmetric_sharded(self.model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, group=self.group)
        else:
            # a direction tracking has already decoded (the new keyframe against the keyframe it was tracked on) is taken
            # over instead of being decoded again: same bits (rows of a batch do not depend on the batch), one row less
            cached, B = {}, len(kf_ii)
            if self.reuse_tracking_decode:
                for e, (a, b) in enumerate(zip(kf_ii, kf_jj)):
                    for row, (src, dst) in ((e, (a, b)), (B + e, (b, a))):
                        pd = getattr(src, "pair_decode", None)
                        if pd is not None and pd[0] == int(dst.frame_id) and a is not b:
                            cached[row] = pd[1]
            res = mast3r_match_symmetric(self.model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, cached=cached or None)
            self.reused_rows += len(cached)
        for kf in kf_ii + kf_jj:          # 46 MB per keyframe at 512x384: consumed (or useless) from here on
            if getattr(kf, "pair_decode", None) is not None:
                kf.pair_decode = None
        return self.add_matched_factors(ii, jj, *res, min_match_frac=min_match_frac, is_reloc=is_reloc)

    @property
    def _driver(self):
        return self.channel is not None and self.channel.is_driver and self.shard_edges

    def _announce_add_factors(self, ii, jj, min_match_frac, is_reloc):
        """Driver: the shards make the same add_factors call; tokens of keyframes they have not seen (or that were
        replaced: a failed relocalisation pops its keyframe) travel first."""
        from mast3r_slam import shard as sh

        ch = self.channel
        with ch.lock:
            new = [int(i) for i in dict.fromkeys(list(ii) + list(jj))
                   if self._sent_tokens.get(int(i)) != int(self.frames[int(i)].frame_id)]
            kf0 = self.frames[int(ii[0])]
            h, w = (int(v) for v in kf0.img_true_shape.reshape(-1)[:2].tolist())
            ints = [len(ii), int(bool(is_reloc)), h, w, int(kf0.feat.shape[1]), len(new)] + [int(i) for i in ii] + \
                   [int(j) for j in jj]
            for i in new:
                ints += [i, int(self.frames[i].frame_id)]
            ch.announce(sh.OP_ADD_FACTORS, ints, [float(min_match_frac)])
            for i in new:
                kf = self.frames[i]
                ch.bcast(kf.feat.contiguous())
                ch.bcast(kf.pos.contiguous())
                self._sent_tokens[i] = int(kf.frame_id)

    def _announce_solve(self, job):
        """Driver: pointmaps / confidences of the keyframes that changed since the shards last got them (rows of the
        job's own copies: exactly what this solve uses), then the solve itself with its start poses."""
        from mast3r_slam import shard as sh

        ch = self.channel
        with ch.lock:
            dirty = job.get("dirty", [])
            hw = int(job["Xs"].shape[1])
            for lo in range(0, len(dirty), sh.POINTMAPS_PER_ANNOUNCEMENT):
                part = dirty[lo:lo + sh.POINTMAPS_PER_ANNOUNCEMENT]
                rows = torch.tensor([r for r, _ in part], device=job["Xs"].device)
                pk = torch.cat((job["Xs"][rows], job["Cs"][rows]), dim=2).contiguous()
                ch.announce(sh.OP_POINTMAPS, [hw, len(part)] + [k for _, k in part])
                ch.bcast(pk)
            kid = {"rays": 0, "calib": 1, "points": 2}[job["kind"]]
            Kf = [float(v) for v in job["K"].reshape(-1).tolist()] if job["K"] is not None else []
            ch.announce(sh.OP_SOLVE, [kid, int(job["pose_data"].shape[0]), int(job["height"]), int(job["width"])], Kf)
            ch.bcast(job["pose_data"])

    def add_matched_factors(self, ii, jj, idx_i2j, idx_j2i, valid_match_j, valid_match_i, Qii, Qjj, Qji, Qij,
                            min_match_frac, is_reloc=False):
        """Second half of add_factors (global_opt.py:56-101): confidence products, match-fraction gate,
        append.  Split out so that precomputed matches can be injected (bench / tests)."""
        batch_inds = torch.arange(idx_i2j.shape[0], device=idx_i2j.device)[:, None].repeat(1, idx_i2j.shape[1])
        Qj = torch.sqrt(Qii[batch_inds, idx_i2j] * Qji)
        Qi = torch.sqrt(Qjj[batch_inds, idx_j2i] * Qij)
        valid_j = valid_match_j & (Qj > self.cfg["Q_conf"])
        valid_i = valid_match_i & (Qi > self.cfg["Q_conf"])
        nj = valid_j.shape[1] * valid_j.shape[2]
        ni = valid_i.shape[1] * valid_i.shape[2]
        match_frac_j = valid_j.sum(dim=(1, 2)) / nj
        match_frac_i = valid_i.sum(dim=(1, 2)) / ni
        ii_t = torch.as_tensor(ii, device=self.device)
        jj_t = torch.as_tensor(jj, device=self.device)
        invalid = torch.minimum(match_frac_j, match_frac_i) < min_match_frac
        invalid = (~(ii_t == (jj_t - 1))) & invalid
        keep = (~invalid).cpu().tolist()          # the ONE host read of the call (a boolean-mask index would be one each)
        if is_reloc and not all(keep):
            return False
        if all(keep):
            sel = lambda t: t
        else:
            rows_kept = torch.tensor([r for r, k in enumerate(keep) if k], dtype=torch.long, device=ii_t.device)
            sel = lambda t: t.index_select(0, rows_kept)
        picked = [("ii", sel(ii_t)), ("jj", sel(jj_t)), ("idx_ii2jj", sel(idx_i2j)), ("idx_jj2ii", sel(idx_j2i)),
                  ("valid_match_j", sel(valid_match_j)), ("valid_match_i", sel(valid_match_i)),
                  ("Q_ii2jj", sel(Qj)), ("Q_jj2ii", sel(Qi))]
        with self.lock:
            for name, rows in picked:
                self._rows[name].append(rows)
        return any(keep)

    ii = property(lambda self: self._rows["ii"].view)
    jj = property(lambda self: self._rows["jj"].view)
    idx_ii2jj = property(lambda self: self._rows["idx_ii2jj"].view)
    idx_jj2ii = property(lambda self: self._rows["idx_jj2ii"].view)
    valid_match_j = property(lambda self: self._rows["valid_match_j"].view)
    valid_match_i = property(lambda self: self._rows["valid_match_i"].view)
    Q_ii2jj = property(lambda self: self._rows["Q_ii2jj"].view)
    Q_jj2ii = property(lambda self: self._rows["Q_jj2ii"].view)

    @property
    def n_edges(self):
        return self._rows["ii"].n

    def two_way_sources(self, n_edges=None):
        """The two-way edge set of prep_two_way_edges WITHOUT the concatenation of the big arrays: (ii, jj) of all 2E
        directed edges (tiny) and the per-edge inputs as the two blocks they are stored in, in the reference's order
        (forward edges, then backward): [(idx, valid, Q) of edges [0, E), (idx, valid, Q) of edges [E, 2E)].
        `n_edges`: only the first n_edges undirected edges (the graph as it stood when they had been added)."""
        v = lambda name: self._rows[name].view if n_edges is None else self._rows[name].view[:n_edges]
        ii = torch.cat((v("ii"), v("jj")), dim=0)
        jj = torch.cat((v("jj"), v("ii")), dim=0)
        return ii, jj, [(v("idx_ii2jj"), v("valid_match_j"), v("Q_ii2jj")), (v("idx_jj2ii"), v("valid_match_i"), v("Q_jj2ii"))]

    def get_unique_kf_idx(self, n_edges=None):
        ii, jj = (self.ii, self.jj) if n_edges is None else (self.ii[:n_edges], self.jj[:n_edges])
        return torch.unique(torch.cat([ii, jj]), sorted=True)

    def prep_two_way_edges(self):
        """global_opt.py:106-112."""
        ii = torch.cat((self.ii, self.jj), dim=0)
        jj = torch.cat((self.jj, self.ii), dim=0)
        idx_ii2jj = torch.cat((self.idx_ii2jj, self.idx_jj2ii), dim=0)
        valid_match = torch.cat((self.valid_match_j, self.valid_match_i), dim=0)
        Q_ii2jj = torch.cat((self.Q_ii2jj, self.Q_jj2ii), dim=0)
        return ii, jj, idx_ii2jj, valid_match, Q_ii2jj

    def get_poses_points(self, unique_kf_idx):
        kfs = [self.frames[idx] for idx in unique_kf_idx]
        Xs = torch.stack([kf.X_canon for kf in kfs])
        T_WCs = Sim3(torch.stack([kf.T_WC.data for kf in kfs]))
        Cs = torch.stack([kf.get_average_conf() for kf in kfs])
        return Xs, T_WCs, Cs

    def _poses_points_cached(self, ids):
        """get_poses_points for a store that stamps its keyframes (KeyframeStore / SharedKeyframes.stamp): the stacked
        pointmaps / average confidences live in a persistent buffer and only the rows whose keyframe changed since the
        last solve are rewritten - the reference (and get_poses_points) restacks all P keyframes per solve (P x 3.1 MB and
        P small launches inside the hand-over section).  The buffer is written and read on the solving stream only, so
        the previous solve's kernels are ordered in front of the next refresh."""
        stamp = self.frames.stamp
        kfs = [self.frames[i] for i in ids]
        P, shp, dev = len(kfs), tuple(kfs[0].X_canon.shape), kfs[0].X_canon.device
        c = self._pm_cache
        if c is None or c["shape"] != shp or c["X"].device != dev or c["X"].shape[0] < P:
            cap = max(64, 2 * P)
            new = dict(shape=shp, X=torch.empty((cap,) + shp, dtype=torch.float32, device=dev),
                       C=torch.empty((cap, shp[0], 1), dtype=torch.float32, device=dev), keys=[None] * cap)
            if c is not None and c["shape"] == shp and c["X"].device == dev:
                n = c["X"].shape[0]
                new["X"][:n], new["C"][:n], new["keys"][:n] = c["X"], c["C"], c["keys"]
            c = self._pm_cache = new
        for r, (i, kf) in enumerate(zip(ids, kfs)):
            key = (int(i), stamp(i))
            if c["keys"][r] != key:
                c["X"][r].copy_(kf.X_canon)
                c["C"][r].copy_(kf.get_average_conf())
                c["keys"][r] = key
                if kf.X_canon.is_cuda:     # the tracking side may replace (and so free) this keyframe's tensors, which
                    cur = torch.cuda.current_stream(dev)   # were allocated on ITS stream, while these copies are queued
                    kf.X_canon.record_stream(cur)
                    kf.C.record_stream(cur)
        T_WCs = Sim3(torch.stack([kf.T_WC.data for kf in kfs]))
        return c["X"][:P], T_WCs, c["C"][:P]

    # ------------------------------------------------------------------
    # The solve in three phases so that a threaded owner (SlamSystem backend="thread") can hold its hand-over lock
    # only around the two short ones: prepare() READS the keyframe store (copies: stack / contiguous), run() works
    # on those copies alone, commit() WRITES the optimised poses back (global_opt.py:145-164 does all three in line).
    def prepare_solve(self, kind, n_edges=None):
        """`n_edges`: solve over the first n_edges (undirected) edges only - the graph of the keyframe task this solve
        belongs to, while a later task's edges may already have been appended by the graph stage."""
        pin = self.cfg["pin"]
        with self.lock:
            if n_edges is not None:
                n_edges = min(int(n_edges), self.n_edges)
            unique_kf_idx = self.get_unique_kf_idx(n_edges)
            ii, jj, sources = self.two_way_sources(n_edges)
        if unique_kf_idx.numel() <= pin:
            self.last_unique_kf_idx = None
            return None
        if ii.is_cuda:      # the lists were allocated (and may be re-allocated) on the graph stage's stream
            cur = torch.cuda.current_stream(ii.device)
            for blk in sources:
                for t in blk:
                    t.record_stream(cur)
        self.last_unique_kf_idx = unique_kf_idx.detach().cpu()
        if hasattr(self.frames, "stamp"):
            Xs, T_WCs, Cs = self._poses_points_cached(self.last_unique_kf_idx.tolist())
        else:
            Xs, T_WCs, Cs = self.get_poses_points(unique_kf_idx)
        K = self.K
        height = width = 0
        if kind == "calib":
            img_size = self.frames[0].img.shape[-2:]
            Xs = constrain_points_to_ray(img_size, Xs, K)
            height, width = int(img_size[0]), int(img_size[1])
        job = dict(kind=kind, pin=pin, unique_kf_idx=unique_kf_idx, unique_kf_idx_host=self.last_unique_kf_idx, K=K,
                   height=height, width=width,
                   pose_data=T_WCs.data[:, 0, :].contiguous(), Xs=Xs.contiguous(), Cs=Cs.contiguous(),
                   edges=(ii, jj, sources))
        if self._driver:     # which rows the shards do not hold in this state (store stamps; no stamps = all of them)
            stamp = getattr(self.frames, "stamp", None)
            dirty = []
            for r, k in enumerate(self.last_unique_kf_idx.tolist()):
                st = stamp(k) if stamp is not None else None
                if st is None or self._sent_stamp.get(k) != st:
                    dirty.append((r, int(k)))
                    self._sent_stamp[k] = st
            job["dirty"] = dirty
        return job

    def run_solve(self, job):
        c, kind, K = self.cfg, job["kind"], job["K"]
        pose_data, Xs, Cs = job["pose_data"], job["Xs"], job["Cs"]
        ii, jj, sources = job["edges"]
        if self._driver:
            self._announce_solve(job)
        sharded = self.shard_edges and torch.distributed.is_available() and torch.distributed.is_initialized()
        gauss_newton_split(kind, pose_data, Xs, Cs, K, ii, jj, sources, c, job["height"], job["width"],
                           group=self.group, sharded=sharded)
        job["Xs"] = job["Cs"] = job["edges"] = None     # the copies are no longer needed

    def commit_solve(self, job):
        pin = job["pin"]
        # the host copy of the indices (made in prepare_solve): a device tensor would cost a synchronisation here
        self.frames.update_T_WCs(Sim3(job["pose_data"][:, None, :])[pin:], job["unique_kf_idx_host"][pin:])

    def _solve(self, kind):
        job = self.prepare_solve(kind)
        if job is not None:
            self.run_solve(job)
            self.commit_solve(job)

    def solve_GN_rays(self):
        """global_opt.py:123-164."""
        self._solve("rays")

    def solve_GN_calib(self):
        """global_opt.py:166-223."""
        self._solve("calib")


def match_symmetric_sharded(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, group=None):
    """mast3r_match_symmetric with the E keyframe pairs partitioned over the ranks (edge_slice): a rank runs the
    two-view forward + matching of its pairs only, then three all-gathers (indices, valid flags, confidences; slices
    padded to the largest) give every rank the full result, in edge order, bit-identical to the one-rank call
    (rows of a batch are computed independently of the batch)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    E = feat_i.shape[0]
    e0, cnt = edge_slice(E, rank, world)
    cmax = (E + world - 1) // world
    dev = feat_i.device
    HW = None
    if cnt > 0:
        sl = slice(e0, e0 + cnt)
        r = mast3r_match_symmetric(model, feat_i[sl], pos_i[sl], feat_j[sl], pos_j[sl], shape_i[sl], shape_j[sl])
        HW = r[0].shape[1]
    hw_t = torch.tensor([HW or 0], device=dev)
    dist.all_reduce(hw_t, op=dist.ReduceOp.MAX, group=group)     # a rank without pairs learns the pixel count
    HW = int(hw_t.item())
    idx = torch.zeros((2, cmax, HW), dtype=torch.long, device=dev)
    val = torch.zeros((2, cmax, HW), dtype=torch.uint8, device=dev)
    q = torch.zeros((4, cmax, HW), dtype=torch.float32, device=dev)
    if cnt > 0:
        idx[0, :cnt], idx[1, :cnt] = r[0], r[1]
        val[0, :cnt], val[1, :cnt] = r[2][..., 0].to(torch.uint8), r[3][..., 0].to(torch.uint8)
        for k in range(4):
            q[k, :cnt] = r[4 + k][..., 0]
    out = []
    for t in (idx, val, q):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        cols = [parts[p][:, :edge_slice(E, p, world)[1]] for p in range(world)]
        out.append(torch.cat(cols, dim=1))
    idx, val, q = out
    return (idx[0], idx[1], val[0].bool()[..., None], val[1].bool()[..., None], q[0][..., None], q[1][..., None],
            q[2][..., None], q[3][..., None])


def broadcast_keyframe(frame, src=0, group=None):
    """A keyframe tracked on rank `src` becomes known to the others: encoder tokens + positions (3.1 MB at 512x384),
    pointmap, confidence and pose - what FactorGraph needs from a keyframe.  In place on every rank."""
    import torch.distributed as dist

    with torch.inference_mode():     # the tokens were produced under inference mode: in-place writes need it too
        for t in (frame.feat, frame.pos, frame.X_canon, frame.C, frame.T_WC.data):
            dist.broadcast(t, src=src, group=group)


def edge_slice(E, rank, world):
    """Contiguous slice of the E directed edges owned by `rank` (balanced to within one edge)."""
    base, rem = divmod(E, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def gauss_newton_sharded(kind, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, cfg, height=0, width=0,
                         group=None):
    """The sharded GN loop on ONE array per per-edge input (the pybind functions' argument form)."""
    return gauss_newton_split(kind, Twc, Xs, Cs, K, ii, jj, [(idx_ii2jj, valid_match, Q)], cfg, height, width,
                              group=group, sharded=True)


def gauss_newton_split(kind, Twc, Xs, Cs, K, ii, jj, sources, cfg, height=0, width=0, group=None, sharded=False):
    """The GN loop of gn_kernels.cu:1181-1225 on the split entry points.  `sources`: the per-edge inputs (idx, valid, Q) of
    the E = len(ii) directed edges as a list of consecutive blocks (FactorGraph.two_way_sources: forward edges, backward
    edges - no concatenation of the [E, HW] arrays).  sharded: the edge kernel runs on this rank's contiguous range of
    the edges; per iteration: local accumulate -> all_reduce(sum) of Hs and gs -> replicated solve + retraction.  Twc is
    updated in place on every rank (identical bits).  Works on any backend torch.distributed offers for the tensors'
    device ("nccl" = RCCL on ROCm).  One rank / not sharded: the same calls without the all-reduce - bit-identical to
    the fused mslam_gauss_newton_* entry points (same kernels in the same order)."""
    rank, world = 0, 1
    if sharded:
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
    P, HW, E = Xs.shape[0], Xs.shape[1], ii.shape[0]
    e0, cnt = edge_slice(E, rank, world)
    dev = Twc.device
    L = _m.lib()
    ws = mast3r_slam_backends._workspace(L.mslam_gn_workspace_bytes(P, E, HW, cnt), dev)
    blocks = torch.zeros(4 * E * 49 + 2 * E * 7, dtype=torch.float32, device=dev)  # one buffer, one all-reduce
    Hs, gs = blocks[: 4 * E * 49], blocks[4 * E * 49:]
    dx = torch.zeros((P - 1, 7), dtype=torch.float32, device=dev)
    kid = {"rays": 0, "calib": 1, "points": 2}[kind]
    sa, sb = {"rays": (cfg["sigma_ray"], cfg["sigma_dist"]), "calib": (cfg["sigma_pixel"], cfg["sigma_depth"]),
              "points": (cfg.get("sigma_point", 0.05), 1.0)}[kind]
    _m.check(L.mslam_gn_begin(_m.ptr(ii), _m.ptr(jj), P, E, HW, _m.ptr(ws), ws.numel(), _m.stream_ptr()), "gn_begin")
    # the pose-independent part (gather, confidence gates) once per call; the iterations stream the result.  The range
    # [e0, e0 + cnt) is compacted block by block of the sources it overlaps
    keep, b0 = [], 0
    for idx_b, vm_b, Q_b in sources:
        nb = idx_b.shape[0]
        lo, hi = max(e0, b0), min(e0 + cnt, b0 + nb)
        if hi > lo:
            idx_l, vm_l, Q_l = (t[lo - b0:hi - b0].contiguous() for t in (idx_b, vm_b, Q_b))   # row slices: views
            keep.append((idx_l, vm_l, Q_l))
            rc = L.mslam_gn_compact_at(_m.ptr(Xs), _m.ptr(Cs), _m.ptr(idx_l), _m.ptr(vm_l), _m.ptr(Q_l), P, HW, E, lo,
                                       hi - lo, lo - e0, cnt, float(cfg["C_conf"]), float(cfg["Q_conf"]), _m.ptr(ws),
                                       ws.numel(), _m.stream_ptr())
            _m.check(rc, "gn_compact_at")
        b0 += nb
    assert b0 == E, (b0, E)
    for _ in range(int(cfg["max_iters"])):
        blocks.zero_()
        rc = L.mslam_gn_accumulate(
            kid, _m.ptr(Twc), _m.ptr(K) if K is not None else 0, P, HW, E, e0, cnt, float(sa), float(sb),
            int(height), int(width), int(cfg["pixel_border"]), float(cfg["depth_eps"]), _m.ptr(Hs), _m.ptr(gs),
            _m.ptr(ws), ws.numel(), _m.stream_ptr())
        _m.check(rc, "gn_accumulate")
        if world > 1:
            dist.all_reduce(blocks, op=dist.ReduceOp.SUM, group=group)
        rc = L.mslam_gn_solve_retract(_m.ptr(Hs), _m.ptr(gs), P, E, HW, _m.ptr(Twc), _m.ptr(dx),
                                      float(cfg["delta_norm"]), _m.ptr(ws), ws.numel(), _m.stream_ptr())
        _m.check(rc, "gn_solve_retract")
    return dx


class _Rows:
    """Row-appendable tensor: `view` = the rows appended so far (a view of a buffer that doubles when full)."""

    def __init__(self, dtype, device):
        self.dtype, self.device, self.buf, self.n = dtype, device, None, 0

    def append(self, rows):
        k = int(rows.shape[0])
        if k == 0:
            return
        if self.buf is None or tuple(self.buf.shape[1:]) != tuple(rows.shape[1:]):
            assert self.n == 0, "row shape changed"
            self.buf = torch.empty((max(8, k),) + tuple(rows.shape[1:]), dtype=self.dtype, device=rows.device)
        if self.n + k > self.buf.shape[0]:
            grown = torch.empty((max(2 * self.buf.shape[0], self.n + k),) + tuple(self.buf.shape[1:]), dtype=self.dtype,
                                device=self.buf.device)
            grown[:self.n] = self.buf[:self.n]
            self.buf = grown
        self.buf[self.n:self.n + k] = rows
        self.n += k

    @property
    def view(self):
        if self.buf is None:
            return torch.as_tensor([], dtype=self.dtype, device=self.device)
        return self.buf[:self.n]
