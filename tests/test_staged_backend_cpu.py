"""Host logic of the staged backend (SlamSystem backend_stages >= 2), on CPU tensors: a solve prepared for keyframe task k
reads exactly the edges the graph held when task k's edges had been appended, although the graph stage may already have
appended task k+1's; the stacked pointmaps / confidences are cached across solves and follow the store's stamps."""
import copy

import numpy as np
import torch

from lietorch_hip import Sim3
from mast3r_slam.frame import Frame, KeyframeStore
from mast3r_slam.global_opt import FactorGraph


def _store(n_kf, hw, seed=0):
    g = torch.Generator().manual_seed(seed)
    store = KeyframeStore()
    for i in range(n_kf):
        f = Frame(i, torch.zeros(1, 3, 4, 4), None, None, None, Sim3.Identity(1))
        f.update_pointmap(torch.rand(hw, 3, generator=g) + 1.0, torch.rand(hw, 1, generator=g) + 0.5)
        store.append(f)
    return store


def _edges(fg, ii, jj, hw, seed):
    g = torch.Generator().manual_seed(seed)
    E = len(ii)
    idx = torch.randint(0, hw, (E, hw), generator=g)
    ones = torch.ones(E, hw, 1, dtype=torch.bool)
    Q = torch.full((E, hw, 1), 4.0)
    assert fg.add_matched_factors(ii, jj, idx, idx.clone(), ones, ones.clone(), Q, Q.clone(), Q.clone(), Q.clone(),
                                  min_match_frac=0.1)


def test_a_solve_sees_the_edges_of_its_own_task_only():
    hw = 32
    store = _store(4, hw)
    fg = FactorGraph(None, store, device="cpu")
    _edges(fg, [0, 0], [1, 2], hw, 1)
    n1 = fg.n_edges
    _edges(fg, [1, 2], [3, 3], hw, 2)          # the graph stage is one task ahead
    assert (n1, fg.n_edges) == (2, 4)
    job = fg.prepare_solve("rays", n_edges=n1)
    ii, jj, sources = job["edges"]
    assert ii.tolist() == [0, 0, 1, 2] and jj.tolist() == [1, 2, 0, 0]                 # forward block, backward block
    assert [tuple(t.shape[0] for t in blk) for blk in sources] == [(2, 2, 2), (2, 2, 2)]
    assert job["unique_kf_idx_host"].tolist() == [0, 1, 2] and job["Xs"].shape[0] == 3   # keyframe 3 is not in this solve
    assert torch.equal(sources[0][0], fg.idx_ii2jj[:n1]) and torch.equal(sources[1][0], fg.idx_jj2ii[:n1])
    full = fg.prepare_solve("rays")
    assert full["edges"][0].shape[0] == 8 and full["unique_kf_idx_host"].tolist() == [0, 1, 2, 3]
    # a limit beyond the graph is the whole graph
    assert fg.prepare_solve("rays", n_edges=99)["edges"][0].shape[0] == 8


def test_stacked_pointmaps_follow_the_store_stamps():
    hw = 32
    store = _store(3, hw)
    fg = FactorGraph(None, store, device="cpu")
    _edges(fg, [0, 1], [1, 2], hw, 3)
    ref = lambda: fg.get_poses_points(torch.arange(3))
    job = fg.prepare_solve("rays")
    Xs, _, Cs = ref()
    assert torch.equal(job["Xs"], Xs) and torch.equal(job["Cs"], Cs)
    keys = list(fg._pm_cache["keys"][:3])
    # the tracking side replaces the newest keyframe (fused pointmap): a new Frame object through __setitem__
    kf = copy.copy(store[2])
    kf.update_pointmap(torch.rand(hw, 3) + 2.0, torch.rand(hw, 1) + 0.5)
    store[2] = kf
    # the local refiner edits an old keyframe IN PLACE and touches it
    store[0].C.add_(0.25)
    store.touch(0)
    job = fg.prepare_solve("rays")
    Xs, _, Cs = ref()
    assert torch.equal(job["Xs"], Xs) and torch.equal(job["Cs"], Cs)
    now = fg._pm_cache["keys"][:3]
    assert now[1] == keys[1] and now[0] != keys[0] and now[2] != keys[2]               # only the changed rows were rewritten
    # growth keeps the rows
    store2 = _store(70, 8, seed=5)
    fg2 = FactorGraph(None, store2, device="cpu")
    _edges(fg2, list(range(69)), list(range(1, 70)), 8, 6)
    job = fg2.prepare_solve("rays", n_edges=10)
    assert job["Xs"].shape[0] == 11
    job = fg2.prepare_solve("rays")
    Xs, _, Cs = fg2.get_poses_points(torch.arange(70))
    assert torch.equal(job["Xs"], Xs) and torch.equal(job["Cs"], Cs)
