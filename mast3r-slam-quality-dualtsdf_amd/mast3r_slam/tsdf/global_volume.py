"""Mirror of mast3r_slam/tsdf/global_volume.py (TSDFVolume, lines 15-140) on a GPU voxel hash.

Same constructor and method names (`integrate`, `query`, `stats`); the python dict
`_voxels[(ix,iy,iz)] -> (tsdf, weight)` becomes an open-addressing table in HBM
(libmslam_hip.so: csrc/tsdf_global.hip).  Inputs may be numpy arrays (as the reference's callers
pass, global_manager.py:101-106) or device tensors; they are moved to the device once.
The per-voxel replay makes results independent of thread scheduling and equal to the reference's
sequential loop; integer keys are bit-exact.  Thread safety: calls are stream-ordered; the
reference's RLock (global_volume.py:30) has no equivalent because there is no host-side state.
"""
import numpy as np
import torch

import mslam_hip as _m


_KEY_BIAS = 1 << 20


def voxel_shard(keys, num_shards):
    """Owner rank of integer voxel keys (n,3): host restatement of the device rule in
    csrc/tsdf_global.hip (pack 3x21-bit biased key, murmur3 finaliser, bits 40.. mod num_shards).
    Used by the multi-rank tests and by callers that route queries to the owning rank."""
    k = np.asarray(keys, np.int64) + _KEY_BIAS
    key = (k[:, 0].astype(np.uint64) << np.uint64(42)) | (k[:, 1].astype(np.uint64) << np.uint64(21)) | k[:, 2].astype(np.uint64)
    with np.errstate(over="ignore"):
        key ^= key >> np.uint64(33); key *= np.uint64(0xff51afd7ed558ccd)
        key ^= key >> np.uint64(33); key *= np.uint64(0xc4ceb9fe1a85ec53)
        key ^= key >> np.uint64(33)
    return ((key >> np.uint64(40)) % np.uint64(num_shards)).astype(np.int64)


class TSDFVolume:
    def __init__(self, voxel_size, truncation, max_weight=100.0, min_weight=1.0e-3, capacity=1 << 22,
                 device="cuda", shard_id=0, num_shards=1, group=None, channel=None):
        self.voxel_size = float(voxel_size)
        self.truncation = float(truncation)
        self.max_weight = float(max_weight)
        self.min_weight = float(min_weight)
        self.capacity = int(capacity)
        self.device = torch.device(device)
        # Voxel sharding (north_star; the reference keeps one dict): this table holds the voxels whose key hashes to
        # `shard_id` of `num_shards` (csrc/tsdf_global.hip: mix64(key) >> 40 mod num_shards).  With num_shards > 1 the
        # methods below are COLLECTIVE over `group`: every rank calls them with the same arguments (integrate keeps only
        # its own voxels of the replicated point list; query all-reduces the owners' look-ups).  `channel`
        # (mast3r_slam/shard.py) is set when one driver rank calls and the others mirror it: the driver announces.
        self.shard_id, self.num_shards = int(shard_id), int(num_shards)
        self.channel = channel
        self.group = channel.group if channel is not None else group
        L = _m.lib()
        nbytes = L.mslam_tsdf_table_bytes(self.capacity)
        if nbytes == 0:
            raise RuntimeError("TSDFVolume: capacity must be a power of two")
        self._table = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws = None
        _m.check(L.mslam_tsdf_table_init(_m.ptr(self._table), nbytes, self.capacity, _m.stream_ptr()), "tsdf_table_init")

    @property
    def _driver(self):
        return self.channel is not None and self.channel.is_driver and self.num_shards > 1

    @property
    def _collective(self):
        return self.num_shards > 1 and (self.group is not None or self.channel is not None)

    def _all_reduce(self, t):
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    # ------------------------------------------------------------------
    def _dev(self, a, dtype):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=self.device, dtype=dtype).contiguous()

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def _header(self):
        out = (np.zeros(8, np.uint32))
        _m.check(_m.lib().mslam_tsdf_header(_m.ptr(self._table), self.capacity, out.ctypes.data, _m.stream_ptr()),
                 "tsdf_header")
        return out

    def maintain(self, max_load=0.5, reserve=0):
        """The reference's dict never fills up; the table does.  One small D2H read (stream sync): raises if samples
        were dropped (table full / coordinate out of the 21-bit key range) and doubles + rehashes the table when more
        than `max_load` of the slots would be taken after `reserve` further insertions (the caller's bound on what it
        integrates before the next call: the table cannot grow in the middle of an integrate).  The synchronous
        pipeline calls it once per backend solve (TSDFGlobalManager.on_after_backend_solve).  Returns (voxels, capacity)."""
        if self._driver:
            from mast3r_slam import shard as sh

            with self.channel.lock:
                self.channel.announce(sh.OP_TSDF_MAINTAIN, [int(reserve)])
        h = self._header()
        if h[1]:
            raise RuntimeError(f"TSDFVolume: samples were dropped (overflow code {int(h[1])}, {int(h[0])} voxels in "
                               f"{self.capacity} slots); raise tsdf_global.hash_capacity")
        while int(h[0]) + int(reserve) > max_load * self.capacity:
            L = _m.lib()
            new_cap = self.capacity * 2
            nbytes = L.mslam_tsdf_table_bytes(new_cap)
            new = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _m.check(L.mslam_tsdf_table_init(_m.ptr(new), nbytes, new_cap, _m.stream_ptr()), "tsdf_table_init")
            _m.check(L.mslam_tsdf_rehash(_m.ptr(self._table), self.capacity, _m.ptr(new), new_cap, _m.stream_ptr()),
                     "tsdf_rehash")
            self._table, self.capacity = new, new_cap
            h = self._header()
            if h[1]:
                raise RuntimeError("TSDFVolume: rehash overflow")
        return int(h[0]), self.capacity

    # ------------------------------------------------------------------
    def integrate(self, points_world, confidences, cam_origin, step_scale=0.5, return_fused=True):
        """global_volume.py:35-72.  Returns the number of fused points (needs one stream sync; pass
        return_fused=False inside a pipeline to stay asynchronous)."""
        pts = self._dev(points_world, torch.float32).reshape(-1, 3)
        n = pts.shape[0]
        if n == 0:
            return 0
        conf = self._dev(confidences, torch.float64).reshape(-1)
        org = self._dev(cam_origin, torch.float32).reshape(3)
        if self._driver:       # the point list is replicated (1.3 MB per 40 000 points), every rank keeps its own voxels
            from mast3r_slam import shard as sh

            pk = torch.empty((n + 1, 4), dtype=torch.float64, device=self.device)
            pk[:n, :3], pk[:n, 3], pk[n, :3], pk[n, 3] = pts, conf, org, 0.0     # f32 -> f64 -> f32 is exact
            with self.channel.lock:
                self.channel.announce(sh.OP_TSDF_FUSE, [n])
                self.channel.bcast(pk)
        L = _m.lib()
        ws = self._workspace(L.mslam_tsdf_integrate_workspace_bytes(n, self.voxel_size, self.truncation, step_scale))
        rc = L.mslam_tsdf_integrate(
            _m.ptr(self._table), self.capacity, _m.ptr(pts), _m.ptr(conf), _m.ptr(org), n, self.voxel_size,
            self.truncation, self.max_weight, float(step_scale), self.shard_id, self.num_shards, _m.ptr(ws),
            ws.numel(), _m.stream_ptr())
        _m.check(rc, "tsdf_integrate")
        if not return_fused:
            return None
        h = self._header()
        if h[1]:
            raise RuntimeError(f"TSDFVolume: voxel table overflow (code {h[1]}); raise capacity (now {self.capacity})")
        if self.num_shards > 1 and self.shard_id != 0:     # the fused-point count is kept by shard 0 (every point once)
            return None
        return int(h[4])

    def query(self, point_world):
        """global_volume.py:93-105 for one point: (tsdf or None, unit gradient (3,) f64 or None)."""
        v, g, st = self.query_batch(np.asarray(point_world, np.float32).reshape(1, 3))
        st = int(st[0])
        if st == 0:
            return None, None
        return float(v[0]), (g[0].cpu().numpy() if st == 2 else None)

    def query_batch(self, points):
        """Vectorised form: (value f64[n], grad f64[n,3], status u8[n]) device tensors;
        status 0 = (None, None), 1 = (value, None), 2 = (value, gradient)."""
        pts = self._dev(points, torch.float32).reshape(-1, 3)
        n = pts.shape[0]
        val = torch.zeros(n, dtype=torch.float64, device=self.device)
        grad = torch.zeros((n, 3), dtype=torch.float64, device=self.device)
        st = torch.zeros(n, dtype=torch.uint8, device=self.device)
        if self._collective:      # owner-computes: every rank looks up what it holds, the sum is the whole answer
            if self._driver:
                from mast3r_slam import shard as sh

                with self.channel.lock:
                    self.channel.announce(sh.OP_TSDF_QUERY, [n])
                    self.channel.bcast(pts)
            lk = self.lookup7(pts)
            rc = _m.lib().mslam_tsdf_query_lookup(_m.ptr(lk), n, self.voxel_size, self.min_weight, _m.ptr(val), _m.ptr(grad),
                                                  _m.ptr(st), _m.stream_ptr())
            _m.check(rc, "tsdf_query_lookup")
            return val, grad, st
        rc = _m.lib().mslam_tsdf_query(_m.ptr(self._table), self.capacity, _m.ptr(pts), n, self.voxel_size,
                                       self.min_weight, _m.ptr(val), _m.ptr(grad), _m.ptr(st), _m.stream_ptr())
        _m.check(rc, "tsdf_query")
        return val, grad, st

    def lookup7(self, pts, pose=None):
        """(state, weight, tsdf) f64[n,7,3] of the seven voxels a query of each point reads, summed over the shards (a voxel
        has one owner: the sum is exact).  `pose` (8,) f32: pts are camera-frame points, moved with it first."""
        n = pts.shape[0]
        lk = torch.zeros((n, 7, 3), dtype=torch.float64, device=self.device)
        rc = _m.lib().mslam_tsdf_lookup7(_m.ptr(self._table), self.capacity, _m.ptr(pts), n, _m.ptr(pose), self.voxel_size,
                                         _m.ptr(lk), _m.stream_ptr())
        _m.check(rc, "tsdf_lookup7")
        return self._all_reduce(lk) if self._collective else lk

    def voxels(self):
        """Dict contents as sorted arrays: keys i64[n,3] (lexicographic), tsdf f64[n], weight f64[n].  Sharded +
        collective: the union over the shards, on every rank (test / export helper: host-side gather)."""
        if self._collective:
            import torch.distributed as dist

            if self._driver:
                from mast3r_slam import shard as sh

                with self.channel.lock:
                    self.channel.announce(sh.OP_TSDF_VOXELS, [])
            mine = self._local_voxels()
            parts = [None] * dist.get_world_size(self.group)
            dist.all_gather_object(parts, mine, group=self.group)
            keys = np.concatenate([p[0] for p in parts]); t = np.concatenate([p[1] for p in parts])
            w = np.concatenate([p[2] for p in parts])
            o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
            return keys[o], t[o], w[o]
        return self._local_voxels()

    def _local_voxels(self):
        n = int(self._header()[0])
        keys = torch.zeros((max(n, 1), 3), dtype=torch.int64, device=self.device)
        t = torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)
        w = torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)
        rc = _m.lib().mslam_tsdf_dump(_m.ptr(self._table), self.capacity, _m.ptr(keys), _m.ptr(t), _m.ptr(w),
                                      max(n, 1), _m.stream_ptr())
        _m.check(rc, "tsdf_dump")
        m = int(self._header()[5])
        keys, t, w = keys[:m].cpu().numpy(), t[:m].cpu().numpy(), w[:m].cpu().numpy()
        o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
        return keys[o], t[o], w[o]

    def stats(self):
        """global_volume.py:136-140."""
        keys, t, w = self.voxels()
        return {"valid_voxels": int((w >= self.min_weight).sum()), "total_voxels": int(len(w))}
