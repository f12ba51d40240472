"""One rank of the two-rank test of the PRODUCT's sharded backend (tests/test_shard_gpu.py): FactorGraph with
shard_edges=True - sharded symmetric pair inference + matching (match_symmetric_sharded), sharded global GN
(gauss_newton_sharded) - on a small room scene, every rank on cuda:0, gloo collectives.  Rank 0 also runs the same
calls un-sharded and writes the comparison.   python tests/shard_worker.py RANK WORLD PORT OUT.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd")]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from lietorch_hip import Sim3
    from mast3r_slam import synthetic
    from mast3r_slam.config import config
    from mast3r_slam.frame import Frame, KeyframeStore
    from mast3r_slam.global_opt import FactorGraph, broadcast_keyframe
    from mast3r_slam.synthetic_gpu import RoomGeometryModel
    import numpy as np

    H, W = 96, 128
    model = RoomGeometryModel(None, dev, H, W)
    ks = [0, 9, 18, 27, 36, 45]

    def build_store(noisy_from_rank0):
        store = KeyframeStore()
        shp = torch.tensor([[H, W]])
        T0 = synthetic.camera_pose(ks[0])
        for i, k in enumerate(ks):
            kt = torch.tensor([float(k)], device=dev)
            fr = Frame(i, model.room.rgb(kt), shp, shp, None)
            fr.feat, fr.pos, _ = model._encode_image(fr.img)
            a, _ = model.room.pair(kt, kt)
            T = synthetic.camera_pose(k).astype(np.float32)
            rel = torch.from_numpy(T).reshape(1, 8).to(dev)
            if i > 0:   # pose noise differs per rank on purpose: rank 0's version is broadcast below
                rel = rel.clone()
                rel[0, :3] += 0.01 * torch.randn(3, device=dev, generator=torch.Generator(device=dev).manual_seed(100 * rank + i))
            fr.T_WC = Sim3(rel)
            fr.update_pointmap(a["pts3d"][0].reshape(-1, 3), a["conf"][0].reshape(-1, 1))
            if noisy_from_rank0:
                broadcast_keyframe(fr, src=0)
            store.append(fr)
        return store

    edges_a = ([0, 1, 2, 0], [1, 2, 3, 2])            # 4 pairs: two per rank
    edges_b = ([3, 1, 4], [4, 4, 5])                  # 3 pairs: ragged slices (2 + 1)
    res = {}
    store = build_store(True)
    fg = FactorGraph(model, store, None, dev, shard_edges=True)
    for ii, jj in (edges_a, edges_b):
        fg.add_factors(ii, jj, config["local_opt"]["min_match_frac"])
    fg.solve_GN_rays()
    poses = torch.stack([store[i].T_WC.data.reshape(8) for i in range(len(ks))])
    state = dict(ii=fg.ii, jj=fg.jj, idx=fg.idx_ii2jj, idx2=fg.idx_jj2ii, vj=fg.valid_match_j, vi=fg.valid_match_i,
                 Q=fg.Q_ii2jj, Q2=fg.Q_jj2ii, poses=poses)
    # every rank must hold the same bits
    same_across_ranks = True
    for k, v in state.items():
        ref = v.clone()
        dist.broadcast(ref, src=0)
        same_across_ranks &= bool(torch.equal(ref, v))
    flag = torch.tensor([int(same_across_ranks)], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        # the same session without sharding (one rank): must be bit-identical
        store_ref = build_store(False)     # rank 0's own poses == what was broadcast
        fg1 = FactorGraph(model, store_ref, None, dev, shard_edges=False)
        for ii, jj in (edges_a, edges_b):
            fg1.add_factors(ii, jj, config["local_opt"]["min_match_frac"])
        fg1.solve_GN_rays()
        poses1 = torch.stack([store_ref[i].T_WC.data.reshape(8) for i in range(len(ks))])
        state1 = dict(ii=fg1.ii, jj=fg1.jj, idx=fg1.idx_ii2jj, idx2=fg1.idx_jj2ii, vj=fg1.valid_match_j,
                      vi=fg1.valid_match_i, Q=fg1.Q_ii2jj, Q2=fg1.Q_jj2ii, poses=poses1)
        res = {k: bool(torch.equal(state[k], state1[k])) for k in state}
        res["same_across_ranks"] = bool(flag.item())
        res["edges"] = int(fg.ii.numel())
        res["pose_moved"] = float((poses1[1:, :3] - torch.stack([torch.from_numpy(synthetic.camera_pose(k)[:3].astype(np.float32)) for k in ks[1:]]).to(dev)).abs().max())
        res["world"] = dist.get_world_size()
    dist.barrier()
    res.update(driver_and_shards(rank, world, dev, model, H, W))
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def driver_and_shards(rank, world, dev, model, H, W):
    """The driver / shard roles of mast3r_slam/shard.py: rank 0 runs the PRODUCT loop (SlamSystem, inline backend) with a
    shard channel - pair inference + matching, the global GN and the global TSDF's voxels (fusion AND the pose
    optimiser's owner-computes queries) sharded over the ranks - while the other ranks serve; then rank 0 repeats the
    session alone.  Every pose, the factor graph and the voxel table must be bit-identical."""
    import numpy as np
    import torch

    from mast3r_slam.config import config
    from mast3r_slam.frame import Frame
    from mast3r_slam.shard import OP_STOP, BackendShard, ShardChannel
    from mast3r_slam.slam_system import SlamSystem
    from mast3r_slam.synthetic_gpu import PoseProximityRetriever

    # the camera path: steady motion, one frame from the other side of the room (tracking is lost there and the session
    # relocalises against the map: main.py:28-71 through the sharded add_factors / solve), steady motion again
    ks = list(range(0, 66, 3)) + [250] + list(range(66, 108, 3))
    n_frames = len(ks)
    saved = config["tracking"]["match_frac_thresh"]
    config["tracking"]["match_frac_thresh"] = 0.72          # a keyframe every ~7 frames at this resolution
    tcfg = dict(config["tsdf_global"], enabled=True, hash_capacity=1 << 16, pre_icp_iters=0, max_iterations=1,
                sync_optimize_per_solve=1, sync_reintegrate_per_solve=2, max_points_per_kf=4000, samples_per_kf=500)
    ch = ShardChannel(dev)

    def session(channel):
        torch.manual_seed(0)
        retr = PoseProximityRetriever(lambda fr: ks[int(fr.frame_id)], 1000)
        system = SlamSystem(model, dev, retriever=retr, frame_group=2, tsdf_global_cfg=tcfg, backend="inline",
                            shard_channel=channel)
        shp = torch.tensor([[H, W]])
        img = model.room.rgb(torch.tensor(ks, device=dev))
        frames = [Frame(j, img[j:j + 1].clone(), shp, shp, None) for j in range(n_frames)]
        results = system.run(frames)
        system.finish()
        vox = system.tsdf_manager.volume.voxels()
        H7, b7, used = system.tsdf_manager.optimizer.normal_equations(
            system.keyframes[1].T_WC.act(system.keyframes[1].X_canon[::97].contiguous()),
            system.keyframes[1].get_average_conf()[::97, 0].contiguous())
        fg = system.factor_graph
        state = dict(poses=torch.stack([r["pose"].reshape(8) for r in results]),
                     kf=torch.stack([system.keyframes[i].T_WC.data.reshape(8) for i in range(len(system.keyframes))]),
                     ii=fg.ii, jj=fg.jj, idx=fg.idx_ii2jj, Q=fg.Q_ii2jj, H7=H7, b7=b7, used=used)
        modes = [int(r["mode"].value) for r in results]
        return system, state, vox, modes

    out = {}
    if rank == 0:
        sys_s, st_s, vox_s, modes_s = session(ch)
        with ch.lock:
            ch.announce(OP_STOP)
        local_voxels = int(sys_s.tsdf_manager.volume._header()[0])
        sys_1, st_1, vox_1, modes_1 = session(None)
        out = {"ds_" + k: bool(torch.equal(st_s[k], st_1[k])) for k in st_s}
        out["ds_modes"] = modes_s == modes_1
        out["ds_voxel_keys"] = bool(np.array_equal(vox_s[0], vox_1[0]))
        out["ds_voxel_values"] = bool(np.array_equal(vox_s[1], vox_1[1]) and np.array_equal(vox_s[2], vox_1[2]))
        out["ds_keyframes"] = len(sys_1.keyframes)
        out["ds_relocalised"] = int(sys_1.stats["relocalised"])
        out["ds_reloc_frames"] = int(sum(m == 2 for m in modes_1))
        out["ds_edges"] = int(sys_1.factor_graph.ii.numel())
        out["ds_voxels"] = int(len(vox_1[0]))
        out["ds_voxels_on_rank0"] = local_voxels
        out["ds_points_used"] = int(st_1["used"].item())
        out["ds_announced"] = {str(k): v for k, v in ch.announced.items()}
    else:
        shard = BackendShard(model, dev, ch, tsdf_global_cfg=tcfg)
        assert shard.serve() == "stop"
    config["tracking"]["match_frac_thresh"] = saved
    return out


if __name__ == "__main__":
    main()
