"""N > 1 path on CPU: two processes over gloo exercise the product's collective layer (tensorrec_amd/sharding.py).
The per-shard compute is supplied by the oracle here (the HIP kernels need a GPU; the same exchange + merge is run
on one GPU with simulated shards in tests/test_gpu_sharding.py).  What is checked: shard bounds tile the items,
all-gathered per-shard top-k lists merge to the exact global top-k, partial rank counts all-reduce to exact ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tensorrec_amd import sharding


def test_shard_bounds_tile_the_items():
    for n_items, world, align in ((1000, 8, 64), (10_000_000, 8, 64), (63, 4, 64), (1, 2, 64), (1682, 3, 32)):
        covered = []
        for r in range(world):
            b, e = sharding.shard_bounds(n_items, world, r, align)
            assert 0 <= b <= e <= n_items and b % align == 0 or b == n_items
            covered.extend(range(b, e)) if n_items < 100000 else covered.append((b, e))
        if n_items < 100000:
            assert covered == list(range(n_items))
        else:
            assert covered[0][0] == 0 and covered[-1][1] == n_items
            assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                      # same data on every rank
        n_users, n_items, d, k = 23, 1000, 16, 10
        u = rng.integers(-2, 3, (n_users, d)).astype(np.float32)       # integer data -> many exact ties
        v = rng.integers(-2, 3, (n_items, d)).astype(np.float32)
        full = O.score_dense_exact(u, v)
        b, e = sharding.shard_bounds(n_items, world, rank, align=64)
        # ---- top-k: local lists carry GLOBAL item ids (item_index_base = b)
        lv, li = O.topk_rows(np.ascontiguousarray(full[:, b:e]), k)
        li = np.where(li >= 0, li + b, -1).astype(np.int32)
        cv, ci = sharding.exchange_topk(torch.from_numpy(lv), torch.from_numpy(li))
        assert cv.shape == (n_users, world * k)
        cv, ci = cv.numpy(), ci.numpy()
        gv, gi = O.topk_rows(full, k)
        for r in range(n_users):                            # oracle merge: (value desc, index asc)
            cand = sorted((-cv[r, j], ci[r, j]) for j in range(world * k) if ci[r, j] >= 0)[:k]
            assert [c[1] for c in cand] == gi[r].tolist()
            assert [-c[0] for c in cand] == gv[r].tolist()
        # ---- ranks: partial counts over the shard's item range, summed by all-reduce
        xu = rng.integers(0, n_users, 200)
        xi = rng.integers(0, n_items, 200)
        tgt = full[xu, xi]
        s = full[xu][:, b:e]
        j = np.arange(b, e)[None, :]
        counts = ((s > tgt[:, None]) | ((s == tgt[:, None]) & (j < xi[:, None]))).sum(1).astype(np.int32)
        total = sharding.reduce_rank_counts(torch.from_numpy(counts.copy()))
        assert np.array_equal(total.numpy() + 1, O.rank_predictions_exact(full)[xu, xi])
        # ---- timing helper
        assert sharding.max_over_ranks(float(rank + 1), "cpu") == float(world)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_and_reduce_over_gloo(world):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def _a2a_worker(rank, world, port, ret):
    """User-partitioned exchanges (sharding.*_a2a): the all-to-all forms must give every rank exactly its users' slice of
    what the all-gather forms give everybody; the merge / k-th-largest kernels are replaced by NumPy here."""
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)
        n_users, n_items, d, k = 29, 1200, 16, 10          # 29 users: not a multiple of the world size (padding path)
        u = rng.integers(-2, 3, (n_users, d)).astype(np.float32)
        v = rng.integers(-2, 3, (n_items, d)).astype(np.float32)
        full = O.score_dense_exact(u, v)
        b, e = sharding.shard_bounds(n_items, world, rank, align=64)
        lv, li = O.topk_rows(np.ascontiguousarray(full[:, b:e]), k)
        li = np.where(li >= 0, li + b, -1).astype(np.int32)

        def merge_np(cv, ci, kk):
            cv, ci = cv.numpy(), ci.numpy()
            ov = np.full((cv.shape[0], kk), -np.inf, np.float32)
            oi = np.full((cv.shape[0], kk), -1, np.int32)
            for r in range(cv.shape[0]):
                cand = sorted((-cv[r, j], ci[r, j]) for j in range(cv.shape[1]) if ci[r, j] >= 0)[:kk]
                for j, (nv, idx) in enumerate(cand):
                    ov[r, j], oi[r, j] = -nv, idx
            return torch.from_numpy(ov), torch.from_numpy(oi)

        def kth_np(table, kk):
            t = np.sort(table.numpy(), axis=0)[::-1]
            return torch.from_numpy(np.ascontiguousarray(t[kk - 1]))

        gv, gi = O.topk_rows(full, k)
        ub, ue, _ = sharding.user_slice(n_users, world, rank)
        mv, mi = sharding.sharded_top_k_a2a(torch.from_numpy(lv), torch.from_numpy(li), k, merge_fn=merge_np)
        assert np.array_equal(mi.numpy(), gi[ub:ue]) and np.array_equal(mv.numpy(), gv[ub:ue])
        rv, ri = sharding.sharded_top_k_a2a(torch.from_numpy(lv), torch.from_numpy(li), k, replicate=True, merge_fn=merge_np)
        assert np.array_equal(ri.numpy(), gi) and np.array_equal(rv.numpy(), gv)
        # floor: k-th largest of all ranks' per-shard top-k "superblock maxima" (here: the shard's k best scores per user)
        sel_max = torch.from_numpy(np.ascontiguousarray(lv.T))                 # [k, n_users]
        floor = sharding.shared_topk_floor_a2a(sel_max, kth_fn=kth_np).numpy()
        gathered = sharding.all_gather_cat(sel_max, dim=0).numpy()             # the all-gather form's table
        assert np.array_equal(floor, np.sort(gathered, axis=0)[::-1][k - 1])
        assert np.array_equal(floor, gv[:, k - 1])                             # = the global k-th best score here
        # item-side maxima of the bf16 filter's bound: MAX all-reduce
        g = sharding.all_reduce_max(torch.tensor([1.0 + rank, 5.0 - rank, 0.5]))
        assert g.tolist() == [float(world), 5.0, 0.5]
        # the known-answer run bench.py starts a multi-GPU launch with (all-gather, all-to-all, all-reduce SUM / MAX)
        assert sharding.a2a_available(torch.zeros(1)) and sharding.collective_selfcheck(torch.device("cpu")) == "ok"
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_user_partitioned_exchanges_over_gloo(world):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_a2a_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g1 = torch.full((5, 3), float(rank + 1))
        g2 = torch.arange(4, dtype=torch.float32) * (rank + 1)
        sharding.all_reduce_sum_([g1, g2])
        assert torch.equal(g1, torch.full((5, 3), float(sum(range(1, world + 1)))))
        assert torch.equal(g2, torch.arange(4, dtype=torch.float32) * sum(range(1, world + 1)))
        assert sharding.all_reduce_scalar(10 + rank, "cpu") == sum(10 + r for r in range(world))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_gradient_all_reduce_helpers_over_gloo():
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
