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
        with open(out, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
