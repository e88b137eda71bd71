"""The cascade's candidate lists: the bf16 refining launches (trec_score_gemm_refine_candidates / _hot, kernel form
LIST of csrc/score_blockmax.hip) also list, per user, every item of a refined (superblock, user) pair whose bf16 score reaches
the provisional floor tauLB - eps, and trec_topk_candidates_finish derives the filter's floor from the listed ITEM scores and
re-scores the survivors exactly -- no table scan, no grouping by superblock, no grouped list kernel.

Bar: values AND item ids bit-identical to the oracle's fp32 restatement of tf.matmul + tf.nn.top_k
(tensorrec/prediction_graphs.py:49-50, tensorrec/recommendation_graphs.py:80), identical to the table-driven tail (tuning
cascade_candidates = 0), and the lists themselves: nothing below the floor, nothing twice, nothing missing that could matter."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tensorrec_amd import ops as _ops, _native
    _native.require_gpu()
    _native.load()
    return _ops


@pytest.fixture
def tuning():
    from tensorrec_amd import _native
    defaults = {"cascade_candidates": 1, "cascade_candidates_cap": 256, "cascade_user_batches": 1, "cascade_prerefine": 1,
                "finish_mixed": 4}

    def set_(name, value):
        assert name in defaults
        _native.set_tuning(name, value)
    yield set_
    for name, value in defaults.items():
        _native.set_tuning(name, value)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(ops, u, v, k, ub=None, ib=None, base=0, **kw):
    dub = dev(ub) if ub is not None else None
    dib = dev(ib) if ib is not None else None
    uop = ops.score_prep_filter(dev(u), sort_users=True, k=k, user_bias=dub)
    iop = ops.score_prep_filter(dev(v), bias=dib, want_gstats=True)
    ops.CANDIDATE_STATS = True                      # (diagnostics: candidates_per_user costs a reduction + a host read)
    try:
        vals, idx = ops.score_topk_filtered(uop, iop, k, dub, dib, item_index_base=base, prefilter="int8", **kw)
    finally:
        ops.CANDIDATE_STATS = False
    return vals.cpu().numpy(), idx.cpu().numpy(), dict(ops.LAST_FILTER_STATS)


@pytest.mark.parametrize("d,biased,n_u,n_i,k", [(128, True, 1300, 300_000, 10), (64, True, 700, 280_011, 16), (128, False, 515, 262_144, 1),
                                                (100, True, 300, 270_000, 5)])
def test_candidate_tail_is_exact_and_equals_the_table_tail(ops, tuning, d, biased, n_u, n_i, k):
    rng = np.random.default_rng(d + n_u)
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.3 * rng.standard_normal(n_u)).astype(np.float32) if biased else None
    ib = (0.3 * rng.standard_normal(n_i)).astype(np.float32) if biased else None
    vals, idx, stats = run(ops, u, v, k, ub, ib, base=1000)
    assert stats["prefilter"] == "int8" and stats.get("tail") == "candidate lists"
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri + 1000) and np.array_equal(vals, rv)
    assert stats["flagged_users"] <= max(1, n_u // 50)
    assert 0.3 * k <= stats["candidates_per_user"] <= 100       # (per row of the padded operand: up to 767 padding rows)
    tuning("cascade_candidates", 0)
    vals0, idx0, stats0 = run(ops, u, v, k, ub, ib, base=1000)
    assert stats0["prefilter"] == "int8" and "tail" not in stats0
    assert np.array_equal(idx0, idx) and np.array_equal(vals0, vals)


@pytest.mark.parametrize("prerefine", [0, 1])
def test_candidate_lists_hold_what_they_must_and_nothing_twice(ops, tuning, prerefine):
    """The lists behind one call, read back: every entry is an item of the catalogue listed once, with a bf16-path score at or
    above the user's provisional floor and within eps of the fp32 score; every item whose fp32 score exceeds the floor by eps
    is listed (it lies in a refined pair: its score is above the k-th largest lower bound).  With the pre-refinement the floor
    rises between the two listing launches: the first one's entries may lie below the FINAL floor (they are only more than needed),
    everything that must be listed still is, and nothing is listed twice (the pre-refined pairs leave the compaction's sight)."""
    tuning("cascade_prerefine", prerefine)
    rng = np.random.default_rng(5)
    n_u, n_i, d, k = 800, 300_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.2 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.2 * rng.standard_normal(n_i)).astype(np.float32)
    dub, dib = dev(ub), dev(ib)
    uop = ops.score_prep_filter(dev(u), sort_users=True, k=k)
    iop = ops.score_prep_filter(dev(v), bias=dib, want_gstats=True)
    ubp = dub.index_select(0, uop.perm)
    n_sb = (n_i + 511) // 512
    table, stride, (rows, overflow), tau8, cands = ops._cascade_stage1(uop, iop, k, ubp, dib, 512, n_sb, 8, None, None, iop.gstats, 0)
    assert not overflow and cands is not None
    n = cands.n.cpu().numpy()
    items = cands.items.cpu().numpy()
    floor0 = cands.floor0.cpu().numpy()
    pad = uop.pad.cpu().numpy() if uop.pad is not None else np.zeros(uop.n, bool)
    assert (n[pad] == 0).all() and (n[~pad] >= k).all() and (n <= cands.cap).all()
    perm = uop.perm.cpu().numpy()
    exact = O.score_dense_exact(u, v, ub, ib)
    st = uop.stats.cpu().numpy()
    g = iop.gstats.cpu().numpy()
    for r in np.flatnonzero(~pad)[::7]:
        ids = items[r, :n[r], 0]
        sh = items[r, :n[r], 1].copy().view(np.float32)
        assert len(np.unique(ids)) == len(ids) and ids.min() >= 0 and ids.max() < n_i
        if not prerefine:
            assert (sh >= floor0[r]).all()
        s = exact[perm[r]]
        eps = st[r, 1] * g[0] * 1.01 + st[r, 0] * g[1] + (d + 2) * 3e-7 * (st[r, 0] * g[0] * 1.01 + abs(ub[perm[r]]) + g[2])
        assert np.abs(sh - s[ids]).max() <= eps * 1.01
        must = np.flatnonzero(s >= floor0[r] + eps * 1.01)
        assert np.isin(must, ids).all()
        assert np.isin(np.argsort(-s, kind="stable")[:k], ids).all()


def test_users_with_more_candidates_than_the_lists_hold_are_redone(ops, tuning):
    """300 copies of the best item: every copy is a candidate of every user who likes it -- more than the 64 entries the lists
    are given here.  Such users are flagged and re-done (wide pass on the table, then fp32); the result stays exact and ties
    come out in index order."""
    rng = np.random.default_rng(9)
    n_u, n_i, d, k = 520, 280_000, 64, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u[:100, 0] += 4.0                                           # 100 users share a taste ...
    star = np.zeros(d, np.float32); star[0] = 20.0              # ... for this item: score ~ 80 against a k-th best of ~ 4.4 * 8
    where = rng.choice(n_i, 300, replace=False)
    v[where] = star
    tuning("cascade_candidates_cap", 64)
    vals, idx, stats = run(ops, u, v, k)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, None, None), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["prefilter"] == "int8" and stats["flagged_users"] >= 10


def test_popular_items_fill_the_wave_queues(ops):
    """Items every user wants (a large bias) sit next to each other: all 128 users of a wave hit in the same 16-item blocks, far
    more than the queue's flush threshold -- the mid-superblock flush must keep every hit (hot superblocks: the dense launch)."""
    rng = np.random.default_rng(21)
    n_u, n_i, d, k = 1100, 300_000, 128, 16
    u = rng.standard_normal((n_u, d)).astype(np.float32) * 0.1
    v = rng.standard_normal((n_i, d)).astype(np.float32) * 0.1
    ib = (0.01 * rng.standard_normal(n_i)).astype(np.float32)
    ib[70_000:70_024] += 5.0                     # 24 adjacent popular items: 16 of them are every user's top-16
    ib[150_003] += 5.0
    vals, idx, stats = run(ops, u, v, k, None, ib)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, None, ib), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["prefilter"] == "int8" and stats["flagged_users"] == 0


def test_user_batches_on_two_streams_equal_one_batch(ops, tuning):
    """The two-stream pipeline (ops.cascade_user_batches: the int8 stage of batch b + 1 next to the refinement and finish of
    batch b), forced on a small problem: 3 batches of a class-sorted operand whose rows have very different scales, some users
    beyond the candidate lists' capacity (re-done at the end, from their batch's table).  Same result as one batch, as the oracle."""
    rng = np.random.default_rng(33)
    n_u, n_i, d, k = 2900, 300_000, 128, 10
    u = (rng.standard_normal((n_u, d)) * np.exp(rng.standard_normal((n_u, 1)))).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u[:120, 0] += 6.0 * np.abs(u[:120]).max(1)
    star = np.zeros(d, np.float32); star[0] = 25.0
    v[rng.choice(n_i, 200, replace=False)] = star
    ub = rng.standard_normal(n_u).astype(np.float32)
    ib = (0.3 * rng.standard_normal(n_i)).astype(np.float32)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    tuning("cascade_candidates_cap", 64)
    old = ops.CASCADE_PIPELINE_MIN_ROWS
    try:
        ops.CASCADE_PIPELINE_MIN_ROWS = 768
        tuning("cascade_user_batches", 3)
        vals, idx, stats = run(ops, u, v, k, ub, ib)
        tuning("cascade_user_batches", 1)
        vals1, idx1, stats1 = run(ops, u, v, k, ub, ib)
    finally:
        ops.CASCADE_PIPELINE_MIN_ROWS = old
    assert stats.get("user_batches") == 3 and "user_batches" not in stats1
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert np.array_equal(idx1, ri) and np.array_equal(vals1, rv)
    # (which users overflow their lists depends on which superblocks are "hot" -- refined for everybody -- and the hot threshold is
    # a fraction of the batch: the counts of the two forms need not agree; both re-do their flagged users exactly)
    assert stats["flagged_users"] >= 50 and stats1["flagged_users"] >= 50
    assert stats["refined_rows"] > 0 and stats1["refined_rows"] > 0      # (hot superblocks count every layout row of their batch)


def test_users_the_int8_bound_says_nothing_about_are_flagged_before_the_lists(ops):
    """Rows 2^-20 of the others share the last scale class with rows a thousand times larger: their int8 images are zero, every
    superblock reaches their threshold.  trec_topk_dense_users flags them before the refining launches (nothing is listed for
    them: no candidate count beyond the capacity), the wide pass / fp32 path re-does them; everything stays exact."""
    rng = np.random.default_rng(77)
    n_u, n_i, d, k = 1500, 300_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    u[::50] *= 2.0 ** -20
    u[1::50] *= 2.0 ** -17
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ops.FILTER_DEBUG = {}
    try:
        vals, idx, stats = run(ops, u, v, k)
        dbg = dict(ops.FILTER_DEBUG)
    finally:
        ops.FILTER_DEBUG = None
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, None, None), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["prefilter"] == "int8" and 30 <= stats["flagged_users"] <= 90
    assert dbg["candidates_q50_90_99_999_max"][-1] <= 256


@pytest.mark.parametrize("kind", ["gauss", "clustered"])
def test_mixed_finish_equals_the_wave_per_user_finish(ops, tuning, kind):
    """trec_topk_candidates_finish_mixed (16 lanes x 1 / 2 / 4 candidates per user -- four users per wave --, users with longer lists
    handed to the wave-per-user form through a device-side list) against trec_topk_candidates_finish (a wave per user for everybody,
    finish_mixed = 0) and the oracle, on Gaussian and on clustered rows (~20-30 candidates per user at this size: with one
    candidate per lane most users go through the hand-over, with four only the longest lists); values and ids bit-identical."""
    rng = np.random.default_rng(17)
    n_u, n_i, d, k = 1500, 300_000, 128, 10
    if kind == "gauss":
        u = rng.standard_normal((n_u, d)).astype(np.float32)
        v = rng.standard_normal((n_i, d)).astype(np.float32)
    else:
        cu, cv = rng.standard_normal((256, d)), rng.standard_normal((256, d))
        u = (cu[rng.integers(0, 256, n_u)] + 0.3 * rng.standard_normal((n_u, d))).astype(np.float32)
        v = (cv[rng.integers(0, 256, n_i)] + 0.3 * rng.standard_normal((n_i, d))).astype(np.float32)
    ub = (0.2 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.2 * rng.standard_normal(n_i)).astype(np.float32)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    cpu = None
    for mixed in (0, 1, 2, 4):
        tuning("finish_mixed", mixed)
        vals, idx, st = run(ops, u, v, k, ub, ib)
        assert st.get("tail") == "candidate lists", st
        assert np.array_equal(idx, ri) and np.array_equal(vals, rv), (kind, mixed)
        cpu = st.get("candidates_per_user") if cpu is None else cpu
    assert cpu > 10, cpu          # (> 16 for most users: finish_mixed = 1 hands nearly everybody over, = 4 the longest lists only)


@pytest.mark.parametrize("n_shards,short", [(8, True), (2, False)])
def test_sixteen_lane_finish_on_item_shards(ops, n_shards, short):
    """finish_lanes=16 (four users per wave) on the lists of item SHARDS that share the all-shard floor, as an N-GPU run makes
    them: with 8 shards a user lists 3-4 candidates per shard and the packed finish answers; with 2 shards most users list
    more than 16 -- they are flagged and re-done on their table column.  Either way the merged lists are the oracle's."""
    from tensorrec_amd import sharding
    rng = np.random.default_rng(3 + n_shards)
    n_u, d, k = 700, 128, 10
    per = 65_536 if short else 160_000
    n_i = per * n_shards
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.2 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.2 * rng.standard_normal(n_i)).astype(np.float32)
    du, dub = dev(u), dev(ub)
    uop = ops.score_prep_filter(du, sort_users=True, k=k, user_bias=dub)
    iops = [ops.score_prep_filter(dev(v[s * per:(s + 1) * per]), bias=dev(ib[s * per:(s + 1) * per]), want_gstats=True)
            for s in range(n_shards)]
    dibs = [dev(ib[s * per:(s + 1) * per]) for s in range(n_shards)]
    gall = torch.stack([io.gstats for io in iops]).max(dim=0).values.contiguous()
    recorded, floor = [], [None]

    def floor_exchange(sel_max):
        if floor[0] is None:
            recorded.append(sel_max.clone())
            return sharding.kth_largest_block_max(sel_max.contiguous(), k)
        return floor[0].clone()

    def stats_exchange(t):
        return gall.clone() if t.numel() == 3 else t
    for s in range(n_shards):
        ops.score_topk_filtered(uop, iops[s], k, dub, dibs[s], item_index_base=s * per, floor_exchange=floor_exchange,
                                stats_exchange=stats_exchange, prefilter="int8")
    floor[0] = sharding.kth_largest_block_max(torch.cat(recorded, dim=0).contiguous(), k)
    lv, li, flagged = [], [], 0
    for s in range(n_shards):
        v_, i_ = ops.score_topk_filtered(uop, iops[s], k, dub, dibs[s], item_index_base=s * per, floor_exchange=floor_exchange,
                                         stats_exchange=stats_exchange, prefilter="int8", finish_lanes=16)
        assert ops.LAST_FILTER_STATS.get("tail") == "candidate lists"
        flagged += ops.LAST_FILTER_STATS["flagged_users"]
        lv.append(v_)
        li.append(i_)
    vals, idx = sharding.merge_topk(torch.cat(lv, dim=1), torch.cat(li, dim=1), k)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    if short:
        assert flagged <= n_u // 20                      # short lists: (nearly) everybody through the packed finish
    else:
        assert flagged > n_u // 2                        # long lists: the flag path did much of the work (over the two shards together)
