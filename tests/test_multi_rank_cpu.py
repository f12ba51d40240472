"""world_size-2 tests of the sharded path on CPU (gloo): the partition rules the GPU ranks use
(global_opt.edge_slice for GN edges, tsdf.global_volume.voxel_shard for voxel ownership) plus the one
exchange step of the path (all-reduce of the zero-padded per-edge Hessian blocks).  The edge kernel and
the voxel integrate are stood in for by the oracle here (no GPU); tests/test_gn_gpu.py and
tests/test_tsdf_gpu.py check the same partitions through the HIP kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from mast3r_slam import synthetic
from mast3r_slam.global_opt import edge_slice
from mast3r_slam.tsdf.global_volume import voxel_shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _graph():
    return synthetic.make_graph(n_kf=5, h=24, w=32, seed=3, stride=6, extra_edges=2, pose_noise=0.01)


def _tsdf_inputs():
    out = []
    for kf in range(2):
        T = synthetic.camera_pose(kf * 10)
        X = synthetic.render_pointmap(T, 48, 64).reshape(-1, 3)
        rng = np.random.default_rng(kf)
        sel = rng.permutation(X.shape[0])[:600]
        out.append((synthetic.sim3_act(T, X[sel]).astype(np.float32), rng.uniform(0.1, 8.0, 600), T[:3].astype(np.float32)))
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = _graph()
        _, ie, je, io, jo = oracle.edge_rows(g["ii"], g["jj"])
        E, P = len(ie), g["Xs"].shape[0]
        e0, cnt = edge_slice(E, rank, world)
        sl = slice(e0, e0 + cnt)
        Hl, gl = oracle.gn_edges("rays", g["Twc"], g["Xs"], g["Cs"], None, ie[sl], je[sl], g["idx_ii2jj"][sl],
                                 g["valid_match"][sl], g["Q"][sl], 0.003, 10.0, 0.0, 1.5)
        # the product's wire format (global_opt.gauss_newton_sharded): one zero-padded buffer, one all-reduce
        blocks = torch.zeros(4 * E * 49 + 2 * E * 7, dtype=torch.float32)
        Hs = blocks[: 4 * E * 49].view(4, E, 7, 7)
        gs = blocks[4 * E * 49:].view(2, E, 7)
        Hs[:, sl] = torch.from_numpy(Hl)
        gs[:, sl] = torch.from_numpy(gl)
        dist.all_reduce(blocks, op=dist.ReduceOp.SUM)
        dx, failed = oracle.gn_solve(Hs.numpy(), gs.numpy(), io, jo, P - 1)
        # voxel ownership: integrate the replicated point list, keep owned voxels only
        vol = oracle.TSDFVolume(0.03, 0.12)
        for pw, conf, org in _tsdf_inputs():
            vol.integrate(pw, conf, org)
        k, t, w = vol.voxels()
        own = voxel_shard(k, world) == rank
        n_own = torch.tensor([int(own.sum())])
        dist.all_reduce(n_own)
        chk = torch.tensor([float(np.sum(t[own] * w[own]))], dtype=torch.float64)
        dist.all_reduce(chk)
        q.put((rank, Hs.numpy().copy(), gs.numpy().copy(), dx, failed, int(n_own), float(chk), k[own]))
    finally:
        dist.destroy_process_group()


def test_edge_slice_partitions():
    for E in (0, 1, 7, 16, 45):
        for world in (1, 2, 3, 8):
            spans = [edge_slice(E, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == E
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_voxel_shard_balanced_and_total():
    rng = np.random.default_rng(0)
    k = rng.integers(-500, 500, (20000, 3))
    for world in (2, 4, 8):
        s = voxel_shard(k, world)
        assert s.min() == 0 and s.max() == world - 1
        cnt = np.bincount(s, minlength=world)
        assert cnt.min() > 0.8 * len(k) / world
    assert (voxel_shard(k, 1) == 0).all()


@pytest.mark.timeout(300)
def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = _graph()
    _, ie, je, io, jo = oracle.edge_rows(g["ii"], g["jj"])
    H_all, g_all = oracle.gn_edges("rays", g["Twc"], g["Xs"], g["Cs"], None, ie, je, g["idx_ii2jj"], g["valid_match"],
                                   g["Q"], 0.003, 10.0, 0.0, 1.5)
    dx_all, failed = oracle.gn_solve(H_all, g_all, io, jo, g["Xs"].shape[0] - 1)
    assert not failed
    for r in res:
        np.testing.assert_array_equal(r[1], H_all)     # disjoint slices + zeros: the sum is exact
        np.testing.assert_array_equal(r[2], g_all)
        np.testing.assert_array_equal(r[3], dx_all)    # replicated solve: identical bits on every rank
        assert not r[4]
    vol = oracle.TSDFVolume(0.03, 0.12)
    for pw, conf, org in _tsdf_inputs():
        vol.integrate(pw, conf, org)
    k, t, w = vol.voxels()
    assert res[0][5] == res[1][5] == len(k)
    np.testing.assert_allclose(res[0][6], float(np.sum(t * w)), rtol=1e-12)
    kk = np.concatenate([res[0][7], res[1][7]])
    o = np.lexsort((kk[:, 2], kk[:, 1], kk[:, 0]))
    np.testing.assert_array_equal(kk[o], k)


def _channel_worker(rank, world, port, q):
    """ShardChannel (mast3r_slam/shard.py) on gloo / CPU: announcements (op code, integer and float arguments, bit-exact
    doubles) and tensor broadcasts arrive on the shard rank in the order the driver issued them."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mast3r_slam.shard import OP_ADD_FACTORS, OP_SOLVE, OP_STOP, ShardChannel

        ch = ShardChannel("cpu")
        got = []
        if ch.is_driver:
            ch.announce(OP_ADD_FACTORS, [3, 0, 96, 128, 48, 1, 0, 1, 2, 1, 2, 3, 3, 17], [0.05])
            ch.bcast(torch.arange(12, dtype=torch.float32).reshape(3, 4))
            ch.announce(OP_SOLVE, [0, 4, 0, 0], [float(np.pi), -1.0e-300, 3.0])
            ch.announce(OP_STOP)
            got = dict(ch.announced)
        else:
            op, ints, floats = ch.receive()
            t = ch.bcast(torch.empty((3, 4), dtype=torch.float32))
            got.append((op, ints, floats, t.tolist()))
            got.append(ch.receive())
            got.append(ch.receive())
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_shard_channel_roundtrip():
    from mast3r_slam.shard import OP_ADD_FACTORS, OP_SOLVE, OP_STOP

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_channel_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == {OP_ADD_FACTORS: 1, OP_SOLVE: 1, OP_STOP: 1}
    a, b, c = res[1]
    assert a[0] == OP_ADD_FACTORS and a[1] == [3, 0, 96, 128, 48, 1, 0, 1, 2, 1, 2, 3, 3, 17] and a[2] == [0.05]
    assert a[3] == torch.arange(12, dtype=torch.float32).reshape(3, 4).tolist()
    assert b == (OP_SOLVE, [0, 4, 0, 0], [float(np.pi), -1.0e-300, 3.0])
    assert c == (OP_STOP, [], [])
