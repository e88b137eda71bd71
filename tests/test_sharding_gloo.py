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


# ---- the gradient exchange of the data-parallel fit (sharding.plan_gradient_exchange and its collectives) --------------------
def _exchange_worker(rank, world, port, ret):
    """Three tables over 3 optimiser steps: A -- rank-disjoint row supports in rank order (identity user features under user
    sharding) -> no exchange, the owner steps its rows; B -- every rank touches every row (the item table) -> reduce-scatter /
    owned-rows Adam / all-gather; C -- small -> all-reduce + the same Adam everywhere.  Against ONE process stepping the summed
    gradients (the oracle's TF-form Adam on both sides): bit-identical at world 2 (a + b commutes), to rounding at world 3."""
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        old_min = sharding.SHARD_MIN_NUMEL
        sharding.SHARD_MIN_NUMEL = 1000
        rows_a, rows_b, d = 101, 67, 16                                  # 67 rows do not divide by 2 or 3: the padded forms
        cuts = [0] + [int(rows_a * (r + 1) / world) - (3 if r + 1 < world else 0) for r in range(world)]   # gaps of untouched rows
        sup_a = (cuts[rank] + (2 if rank else 0), cuts[rank + 1])
        names = ["A", "B", "C"]
        shapes = {"A": (rows_a, d), "B": (rows_b, d), "C": (7, 1)}
        plan = sharding.plan_gradient_exchange(names, shapes, {"A": sup_a}, "cpu")
        assert plan.mode == {"A": "disjoint", "B": "sharded", "C": "replicated"}
        assert plan.bounds["A"][0] == 0 and plan.bounds["A"][-1] == rows_a and plan.own["A"][0] <= sup_a[0] and sup_a[1] <= plan.own["A"][1]
        assert plan.bounds["B"][-1] == rows_b and len(plan.bounds["B"]) == world + 1
        wire = plan.wire_bytes_per_step(shapes)
        assert wire["A"] == 0 and wire["B"] == 2.0 * (world - 1) / world * rows_b * d * 4
        # overlapping or out-of-order supports are NOT disjoint
        p2 = sharding.plan_gradient_exchange(["A"], shapes, {"A": (0, rows_a)}, "cpu")
        assert p2.mode["A"] == "sharded"
        p3 = sharding.plan_gradient_exchange(["A"], shapes, {"A": (cuts[world - 1 - rank], cuts[world - rank])}, "cpu")
        assert p3.mode["A"] == ("sharded" if world > 1 else "disjoint")
        rng = np.random.default_rng(5)                                   # same stream on every rank: all ranks' gradients known
        w = {n: rng.standard_normal(shapes[n]).astype(np.float32) for n in names}
        ref = {n: (w[n].copy(), np.zeros(shapes[n], np.float32), np.zeros(shapes[n], np.float32)) for n in names}
        mine = {n: (torch.from_numpy(w[n].copy()), torch.zeros(shapes[n]), torch.zeros(shapes[n])) for n in names}
        for step in range(3):
            grads = []
            for r in range(world):
                g = {n: rng.standard_normal(shapes[n]).astype(np.float32) for n in names}
                lo, hi = cuts[r] + (2 if r else 0), cuts[r + 1]
                g["A"][:lo] = 0
                g["A"][hi:] = 0
                grads.append(g)
            lr_t = 0.05 * (step + 1)
            for n in names:                                              # the single process: sum in rank order, dense Adam
                total = grads[0][n].copy()
                for r in range(1, world):
                    total = total + grads[r][n]
                O.adam_tf_step(ref[n][0], ref[n][1], ref[n][2], total, lr_t)
            g_mine = {n: torch.from_numpy(grads[rank][n].copy()) for n in names}
            own_b, work = sharding.reduce_scatter_rows(g_mine["B"], plan.bounds["B"], rank, async_op=True)
            work_c = dist.all_reduce(g_mine["C"], async_op=True)
            lo, hi = plan.own["A"]
            wa, ma, va = (t.numpy()[lo:hi] for t in mine["A"])
            O.adam_tf_step(wa, ma, va, g_mine["A"].numpy()[lo:hi], lr_t)
            work_c.wait()
            O.adam_tf_step(*(t.numpy() for t in mine["C"]), g_mine["C"].numpy(), lr_t)
            if work is not None:
                work.wait()
            lo, hi = plan.own["B"]
            wb, mb, vb = (t.numpy()[lo:hi] for t in mine["B"])
            O.adam_tf_step(wb, mb, vb, own_b.numpy(), lr_t)
            sharding.all_gather_rows(mine["B"][0], plan.bounds["B"], rank)
        # before the sync: the owned rows are right, A's other rows are stale
        lo, hi = plan.own["A"]
        exact = world == 2
        same = (lambda a, b: np.array_equal(a, b)) if exact else (lambda a, b: np.allclose(a, b, rtol=1e-5, atol=1e-6))
        assert np.array_equal(mine["A"][0].numpy()[lo:hi], ref["A"][0][lo:hi])          # (disjoint rows: one contributor, always exact)
        assert same(mine["B"][0].numpy(), ref["B"][0]) and same(mine["C"][0].numpy(), ref["C"][0])
        sharding.sync_owned_rows(list(mine["A"]), plan.bounds["A"])
        sharding.sync_owned_rows(list(mine["B"][1:]), plan.bounds["B"])
        for n in names:
            for j in range(3):
                assert same(mine[n][j].numpy(), ref[n][j]), (n, j)
        assert np.array_equal(mine["A"][0].numpy(), ref["A"][0])
        sharding.SHARD_MIN_NUMEL = old_min
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gradient_exchange_plan_and_step_equivalence_over_gloo(world):
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}
