"""Exact ranks at scale (VERDICT r1 "missing" #2):

  * trec_rank_rows_chunked -- full rows of more than 32768 items: sorted 32768-item chunks + cross-chunk binary
    searches, bit-exact against the oracle's double-top_k restatement (recommendation_graphs.py:73-82), ties / +-0 / +-inf
    / heavy-tie rows included, and 4,096 x 200,000 ranks in well under a second;
  * trec_score_gemm_rankcount -- ranks of chosen pairs as the epilogue of the fp32 MFMA score kernel: no [U, I] slab;
    bit-exact against O.rank_predictions_exact of the oracle's fp32 score matrix; item ranges (shards) add."""
import time

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tensorrec_amd import ops as _ops, _native
    _native.require_gpu()
    _native.load()
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------ chunked full-row ranks
@pytest.mark.parametrize("n_items,case", [(100_003, "random"), (65_536, "random"), (32_769, "random"),
                                          (100_003, "some_ties"), (70_000, "heavy_ties"), (98_304, "special")])
def test_rank_rows_chunked_bit_exact(ops, n_items, case):
    rng = np.random.default_rng(n_items % 1000 + len(case))
    n_users = 23
    s = rng.standard_normal((n_users, n_items)).astype(np.float32)
    if case == "some_ties":                                   # ~1500 tied items per row, spread over the chunks
        for u in range(n_users):
            src = rng.integers(0, n_items, 750)
            dst = rng.integers(0, n_items, 750)
            s[u, dst] = s[u, src]
    elif case == "heavy_ties":                                # integer-valued scores: thousands of ties per chunk
        s[::2] = np.round(s[::2] * 3)
    elif case == "special":
        s[:, 5] = np.inf
        s[:, 40_000] = np.inf
        s[:, 77] = -np.inf
        s[:, 90_001] = -np.inf
        s[:, 100:200:2] = 0.0
        s[:, 50_000:50_100:2] = -0.0
    got = ops.rank_rows(dev(s)).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, O.rank_predictions_exact(s))


def test_rank_rows_4096_by_200000_under_a_second(ops):
    n_users, n_items = 4096, 200_000
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    s = torch.randn((n_users, n_items), device="cuda", generator=gen)
    ops.rank_rows(s[:64])                                     # warm-up (kernel attributes, workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ops.rank_rows(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("rank_rows %d x %d: %.3f s on the device" % (n_users, n_items, dt))
    assert dt < 1.0
    sub = np.arange(0, n_users, 512)
    assert np.array_equal(r[sub].cpu().numpy(), O.rank_predictions_exact(s[sub].cpu().numpy()))


# ------------------------------------------------------------------------------------------------ fused rank counts
def _pairs(rng, n_users, n_items, max_per_user):
    per_user = rng.integers(0, max_per_user + 1, n_users)
    per_user[0] = 0
    per_user[1] = max_per_user
    cols = [np.sort(rng.choice(n_items, size=p, replace=False)) for p in per_user]
    indptr = np.concatenate([[0], np.cumsum(per_user)]).astype(np.int64)
    return indptr, np.concatenate(cols).astype(np.int32), np.repeat(np.arange(n_users), per_user)


@pytest.mark.parametrize("d,biased,integer", [(128, True, False), (100, True, False), (64, False, False),
                                              (32, True, True), (256, True, False)])
def test_rankcount_fused_bit_exact_vs_oracle(ops, d, biased, integer):
    rng = np.random.default_rng(d + biased)
    n_users, n_items = 150, 40_000 + 13
    if integer:                                               # exact ties everywhere: the index tie-break decides
        u = rng.integers(-2, 3, (n_users, d)).astype(np.float32)
        v = rng.integers(-2, 3, (n_items, d)).astype(np.float32)
    else:
        u = rng.standard_normal((n_users, d)).astype(np.float32)
        v = rng.standard_normal((n_items, d)).astype(np.float32)
    ub = rng.standard_normal(n_users).astype(np.float32) if biased else None
    ib = rng.standard_normal(n_items).astype(np.float32) if biased else None
    if integer and biased:
        ub, ib = np.round(ub), np.round(ib)
    indptr, xi, xu = _pairs(rng, n_users, n_items, 75)        # up to 75 targets per user: three resident rows
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    dub = dev(ub) if biased else None
    dib = dev(ib) if biased else None
    counts = ops.rank_counts_fused(u_op, v_op, kpad, d, indptr, dev(xi), dub, dib).cpu().numpy()
    ref = O.rank_predictions_exact(O.score_dense_exact(u, v, ub, ib))
    assert np.array_equal(counts + 1, ref[xu, xi])
    # item ranges add: two shards with global item ids; the targets' exact scores travel with the pairs
    cut = 17_024
    tgt = ops.pair_scores_exact(u_op, v_op, kpad, d, dev(xu.astype(np.int32)), dev(xi), dub, dib)
    c0 = ops.rank_counts_fused(u_op, v_op[:cut].contiguous(), kpad, d, indptr, dev(xi), dub,
                               dib[:cut].contiguous() if biased else None, n_chunks=3, target_scores=tgt)
    c1 = ops.rank_counts_fused(u_op, v_op[cut:].contiguous(), kpad, d, indptr, dev(xi), dub,
                               dib[cut:].contiguous() if biased else None, item_index_base=cut, n_chunks=5,
                               target_scores=tgt)
    assert np.array_equal((c0 + c1).cpu().numpy(), counts)


def test_rankcount_targets_scoring_minus_infinity(ops):
    """Items masked out with a -inf bias score -inf for every user; a TARGET among them ties with all the others and the
    lower-index ones precede it (tf.nn.top_k's order) -- also in item blocks wholly below the target's index, where the kernel
    counts s >= t as s > float_pred(t) and float_pred(-inf) = -inf."""
    rng = np.random.default_rng(17)
    n_users, n_items, d = 64, 20_000, 64
    u = rng.standard_normal((n_users, d)).astype(np.float32)
    v = rng.standard_normal((n_items, d)).astype(np.float32)
    ub = rng.standard_normal(n_users).astype(np.float32)
    ib = rng.standard_normal(n_items).astype(np.float32)
    masked = np.sort(rng.permutation(n_items)[:400])
    ib[masked] = -np.inf
    per_user = 6
    xi = np.stack([np.concatenate([rng.permutation(masked)[:3], rng.permutation(n_items)[:3]]) for _ in range(n_users)])
    xi = np.sort(xi, axis=1).reshape(-1).astype(np.int32)
    xu = np.repeat(np.arange(n_users), per_user)
    indptr = np.arange(0, (n_users + 1) * per_user, per_user, dtype=np.int64)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    counts = ops.rank_counts_fused(u_op, v_op, kpad, d, indptr, dev(xi), dev(ub), dev(ib)).cpu().numpy()
    ref = O.rank_predictions_exact(O.score_dense_exact(u, v, ub, ib))
    assert np.array_equal(counts + 1, ref[xu, xi])


def test_rankcount_matches_slab_path_for_cosine_and_euclidean(ops):
    """cosine / Euclidean scores have no bit-exact oracle (normalisation / sqrt rounding); the fused counts must equal the
    ranks of the score matrix the STORE kernel writes from the same operands (what predict_rank ranks)."""
    rng = np.random.default_rng(5)
    n_users, n_items, d = 96, 33_000, 64
    u = rng.standard_normal((n_users, d)).astype(np.float32)
    v = rng.standard_normal((n_items, d)).astype(np.float32)
    ub, ib = rng.standard_normal(n_users).astype(np.float32), rng.standard_normal(n_items).astype(np.float32)
    indptr, xi, xu = _pairs(rng, n_users, n_items, 40)
    for mode, normalize in ((ops.MODE_DOT, True), (ops.MODE_EUCLIDEAN, False)):
        want_sq = mode == ops.MODE_EUCLIDEAN
        u_op, u_sq, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, normalize=normalize, want_sqnorm=want_sq)
        v_op, v_sq, _ = ops.score_prep(dev(v), ops.DTYPE_F32, normalize=normalize, want_sqnorm=want_sq)
        slab = ops.score_store(u_op, v_op, ops.DTYPE_F32, kpad, dev(ub), dev(ib), mode, u_sq, v_sq).cpu().numpy()
        counts = ops.rank_counts_fused(u_op, v_op, kpad, d, indptr, dev(xi), dev(ub), dev(ib), mode, u_sq, v_sq)
        assert np.array_equal(counts.cpu().numpy() + 1, O.rank_predictions_exact(slab)[xu, xi])


def test_predict_rank_of_interactions_at_1m_items_without_a_slab(ops):
    """The public method at a catalogue where a [users, items] slab would be 8 GB per 2048 users: ranks of the test
    interactions, bit-exact against the oracle on a sampled user tile."""
    import tensorrec_amd as T
    rng = np.random.default_rng(9)
    n_users, n_items, d = 2048, 1_000_000, 64
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    cols = rng.integers(0, n_items, size=(n_users, 20))
    inter = sp.csr_matrix((np.ones(n_users * 20, np.float32), cols.reshape(-1), np.arange(0, n_users * 20 + 1, 20)),
                          shape=(n_users, n_items))
    inter.sum_duplicates()
    inter.data[:] = 1.0
    model = T.TensorRec(n_components=d, seed=2)
    model.fit(inter, uf, itf, epochs=1)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    pr = model.predict_rank_of_interactions(uf, itf, inter)
    dt = time.perf_counter() - t0
    peak = torch.cuda.max_memory_allocated() - base
    print("predict_rank_of_interactions %d users x %d items, %d pairs: %.3f s, peak extra device memory %.2f GB"
          % (n_users, n_items, inter.nnz, dt, peak / 2 ** 30))
    assert peak < 2 * 2 ** 30                                  # a slab for 2048 users alone would be 8 GB
    w = model.get_weights()
    sample = np.arange(0, n_users, 64)
    us = O.spmm_exact(uf[sample], w["linear_weights_user_0"])
    vs = O.spmm_exact(itf, w["linear_weights_item"])
    ub = O.spmm_exact(uf[sample], w["user_feature_biases"]).reshape(-1)
    ib = O.spmm_exact(itf, w["item_feature_biases"]).reshape(-1)
    ref = O.rank_predictions_exact(O.score_dense_exact(us, vs, ub, ib))
    m = sp.csr_matrix(inter)
    m.sort_indices()
    for k, uidx in enumerate(sample):
        sel = pr.rows == uidx
        c = m.indices[m.indptr[uidx]:m.indptr[uidx + 1]]
        assert np.array_equal(pr.ranks[sel], ref[k, c])
