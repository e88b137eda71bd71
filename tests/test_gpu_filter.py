"""K2f -- the exact fp32 top-k through the bf16 filter (csrc/topk_filter.hip, ops.score_topk_filtered).

Bar: values AND item ids bit-identical to the oracle's fp32 restatement of tf.matmul + tf.nn.top_k
(tensorrec/prediction_graphs.py:49-50, tensorrec/recommendation_graphs.py:80: oracle/tr_oracle.c:orc_score_dense +
O.topk_rows) and to the fp32 MFMA path, on random, tied, near-tied and adversarially aligned inputs; the proven bound
eps_u must dominate every observed |bf16 score - fp32 score|."""
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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_filtered(ops, u, v, k, ub=None, ib=None, normalize=False, **kw):
    du, dv = dev(u), dev(v)
    dub = dev(ub) if ub is not None else None
    dib = dev(ib) if ib is not None else None
    uop = ops.score_prep_filter(du, normalize=normalize)
    iop = ops.score_prep_filter(dv, normalize=normalize, bias=dib, want_gstats=True)
    vals, idx = ops.score_topk_filtered(uop, iop, k, dub, dib, **kw)
    return vals.cpu().numpy(), idx.cpu().numpy(), dict(ops.LAST_FILTER_STATS), uop, iop


def exact_reference(u, v, k, ub=None, ib=None):
    return O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)


@pytest.mark.parametrize("d,biased", [(128, True), (128, False), (64, True), (100, True), (32, False), (256, True)])
def test_filtered_topk_bit_exact_vs_oracle(ops, d, biased):
    rng = np.random.default_rng(d + biased)
    n_u, n_i, k = 700, 40000 + 77, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ub = (0.1 * rng.standard_normal(n_u)).astype(np.float32) if biased else None
    ib = (0.1 * rng.standard_normal(n_i)).astype(np.float32) if biased else None
    vals, idx, stats, _, _ = run_filtered(ops, u, v, k, ub, ib)
    rv, ri = exact_reference(u, v, k, ub, ib)
    assert np.array_equal(idx, ri)
    assert np.array_equal(vals, rv)
    assert stats["flagged_users"] <= n_u // 20           # the filter itself must carry (almost) everybody


def test_filtered_equals_fp32_mfma_path_cosine_unnormalised_inputs(ops):
    """cosine: operands are normalised by the prep kernel exactly as trec_score_prep does, so the filtered result must
    equal the fp32 two-stage path bit for bit (and the oracle's cosine scores within 1e-6 -- l2-normalise rounding)."""
    rng = np.random.default_rng(5)
    n_u, n_i, d, k = 300, 50000, 128, 10
    u = (rng.standard_normal((n_u, d)) * rng.uniform(0.1, 30, (n_u, 1))).astype(np.float32)
    v = (rng.standard_normal((n_i, d)) * rng.uniform(0.1, 30, (n_i, 1))).astype(np.float32)
    vals, idx, stats, uop, iop = run_filtered(ops, u, v, k, normalize=True)
    u32, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, normalize=True)
    v32, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32, normalize=True)
    assert torch.equal(u32, uop.f32) and torch.equal(v32, iop.f32)
    vb, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16, normalize=True)
    assert torch.equal(vb, iop.bf16)
    ev, ei = ops.score_topk(u32, v32, ops.DTYPE_F32, kpad, k, method="two_stage")
    assert np.array_equal(idx, ei.cpu().numpy()) and np.array_equal(vals, ev.cpu().numpy())
    ref = O.cosine_dense(u, v)
    np.testing.assert_allclose(vals, np.take_along_axis(ref, idx.astype(np.int64), 1), rtol=0, atol=1e-6)


def _eps_of(ops, uop, iop, ub, kpad):
    """eps_u as the floor kernel computes it: floor = pred(pred(0 - 2 eps))."""
    n_u = uop.n
    tau = torch.zeros((n_u,), dtype=torch.float32, device="cuda")
    floor = torch.empty((n_u,), dtype=torch.float32, device="cuda")
    flag = torch.empty((n_u,), dtype=torch.int32, device="cuda")
    nf = torch.zeros((1,), dtype=torch.int32, device="cuda")
    N = ops.N
    N.call("trec_topk_filter_floor", N.ptr(tau), N.ptr(uop.stats), N.ptr(ub), N.ptr(iop.gstats), kpad, n_u, N.ptr(floor),
           N.ptr(flag), N.ptr(nf))
    return (-floor / 2).cpu().numpy()


@pytest.mark.parametrize("case", ["random", "aligned_rounding", "large_magnitude", "biased"])
def test_error_bound_dominates_observed_error(ops, case):
    """max |bf16-path score - fp32 score| / eps_u <= 1 on every pair; 'aligned_rounding' makes the rounding errors of the
    user row parallel to the item rows, the case in which the Cauchy-Schwarz step is tight."""
    rng = np.random.default_rng(11)
    n_u, n_i, d = 256, 4096, 128
    ub = ib = None
    if case == "aligned_rounding":
        # every component rounds DOWN by almost half a bf16 ulp; items are the all-ones direction: <dx, y> = |dx||y|
        u = (np.float32(2.0) ** rng.integers(-1, 3, (n_u, 1)).astype(np.float32)) * \
            np.full((n_u, d), 1.0 + 2.0 ** -8 - 2.0 ** -20, np.float32)
        v = np.ones((n_i, d), np.float32) * (np.float32(2.0) ** rng.integers(-2, 3, (n_i, 1)).astype(np.float32))
    elif case == "large_magnitude":
        u = (rng.standard_normal((n_u, d)) * 1e3).astype(np.float32)
        v = (rng.standard_normal((n_i, d)) * 1e2).astype(np.float32)
    else:
        u = rng.standard_normal((n_u, d)).astype(np.float32)
        v = rng.standard_normal((n_i, d)).astype(np.float32)
    if case == "biased":
        ub = (3 * rng.standard_normal(n_u)).astype(np.float32)
        ib = (3 * rng.standard_normal(n_i)).astype(np.float32)
    dub = dev(ub) if ub is not None else None
    dib = dev(ib) if ib is not None else None
    uop = ops.score_prep_filter(dev(u))
    iop = ops.score_prep_filter(dev(v), bias=dib, want_gstats=True)
    kpad = uop.kpad
    eps = _eps_of(ops, uop, iop, dub, kpad)
    s_bf = ops.score_store(uop.bf16, iop.bf16, ops.DTYPE_BF16, kpad, dub, dib).cpu().numpy()
    s_32 = O.score_dense_exact(u, v, ub, ib)
    err = np.abs(s_bf.astype(np.float64) - s_32.astype(np.float64))
    ratio = (err / eps[:, None]).max()
    print("case %s: max |err| / eps = %.3f (eps mean %.3g)" % (case, ratio, eps.mean()))
    assert ratio <= 1.0
    if case == "aligned_rounding":
        assert ratio > 0.5                  # the bound is not slack by construction in its tight case


def test_near_ties_within_the_bf16_error(ops):
    """Scores spaced far below 2^-9 relative: bf16 cannot order them, the filter must keep them all and the fp32
    re-scoring must produce the exact order (ties by index).  40 near-copies of each user's best item are scattered
    over distinct superblocks."""
    rng = np.random.default_rng(3)
    n_u, n_i, d, k = 128, 60000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = (0.3 * rng.standard_normal((n_i, d))).astype(np.float32)
    # items aligned with a few users, perturbed at the 1e-5 level, placed in different superblocks
    for j, uu in enumerate(range(0, 24)):
        slots = (np.arange(20) * 2557 + 31 * j) % n_i
        v[slots] = u[uu] * 2.0 + (1e-5 * rng.standard_normal((20, d))).astype(np.float32)
    vals, idx, stats, _, _ = run_filtered(ops, u, v, k)
    rv, ri = exact_reference(u, v, k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["flagged_users"] < n_u             # handled by the filter, not only by the fallback


def test_exact_ties_keep_index_order(ops):
    """Small-integer embeddings are exact in bf16: thousands of exactly tied scores; order = (value desc, index asc)."""
    rng = np.random.default_rng(4)
    n_u, n_i, d, k = 200, 33000, 64, 10
    u = rng.integers(-2, 3, (n_u, d)).astype(np.float32)
    v = rng.integers(-2, 3, (n_i, d)).astype(np.float32)
    vals, idx, stats, _, _ = run_filtered(ops, u, v, k)
    rv, ri = exact_reference(u, v, k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def test_degenerate_all_scores_within_the_bound_falls_back_exactly(ops):
    """Every item within 2 eps of the best (near-identical item rows): no filter can certify anything -- every user is
    flagged and re-done on the fp32 MFMA path; the result is still the oracle's."""
    rng = np.random.default_rng(6)
    n_u, n_i, d, k = 96, 20000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    base = rng.standard_normal((1, d)).astype(np.float32)
    v = base + (1e-6 * rng.standard_normal((n_i, d))).astype(np.float32)
    vals, idx, stats, _, _ = run_filtered(ops, u, v, k)
    rv, ri = exact_reference(u, v, k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["flagged_users"] == n_u


def test_full_list_in_one_superblock_is_flagged_not_truncated(ops):
    """More near-tied items inside ONE (superblock, half-wave) than a stage-3 list holds: the user must be flagged (a
    full list may have dropped survivors) and come out exact."""
    rng = np.random.default_rng(7)
    n_u, n_i, d, k = 64, 30000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = (0.2 * rng.standard_normal((n_i, d))).astype(np.float32)
    v[1024:1024 + 40] = u[5] * 3.0 + (1e-6 * rng.standard_normal((40, d))).astype(np.float32)   # 40 survivors, one block
    vals, idx, stats, _, _ = run_filtered(ops, u, v, k)
    rv, ri = exact_reference(u, v, k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["flagged_users"] >= 1


def test_item_shards_with_shared_floor_and_stats(ops):
    """Two item shards on one GPU: tau = k-th largest maximum over both shards, item maxima all-reduced (MAX); the merge
    of the per-shard lists is the whole-catalogue exact top-k."""
    rng = np.random.default_rng(8)
    n_u, n_i, d, k = 300, 70000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = rng.standard_normal(n_u).astype(np.float32)
    ib = rng.standard_normal(n_i).astype(np.float32)
    cut = 36864
    du, dub = dev(u), dev(ub)
    shards = [(0, cut), (cut, n_i)]
    iops = [ops.score_prep_filter(dev(v[a:b]), bias=dev(ib[a:b]), want_gstats=True) for a, b in shards]
    gmax = torch.maximum(iops[0].gstats, iops[1].gstats)
    # pass 1: every shard's k largest maxima (what sharding.shared_topk_floor all-gathers)
    captured = []

    def capture(sel_max):
        captured.append(sel_max.clone())
        return torch.full((n_u,), -np.inf, device="cuda")

    uop = ops.score_prep_filter(du)
    for (a, b), iop in zip(shards, iops):
        ops.score_topk_filtered(uop, iop, k, dub, dev(ib[a:b]), item_index_base=a, floor_exchange=capture,
                                stats_exchange=lambda g: gmax)
    both = torch.cat(captured, dim=0)                                   # [2k, n_u]
    tau_glob = torch.sort(both, dim=0, descending=True).values[k - 1].contiguous()
    parts_v, parts_i = [], []
    for (a, b), iop in zip(shards, iops):
        pv, pi = ops.score_topk_filtered(uop, iop, k, dub, dev(ib[a:b]), item_index_base=a,
                                         floor_exchange=lambda sm: tau_glob, stats_exchange=lambda g: gmax)
        parts_v.append(pv)
        parts_i.append(pi)
    mv, mi = ops.topk_merge(torch.cat(parts_v, 1).contiguous(), torch.cat(parts_i, 1).contiguous(), k)
    rv, ri = exact_reference(u, v, k, ub, ib)
    assert np.array_equal(mi.cpu().numpy(), ri) and np.array_equal(mv.cpu().numpy(), rv)


def test_model_predict_top_k_uses_the_filter_and_is_exact(ops):
    """Through the public API: precision='fp32' top-k on a catalogue above the two-stage threshold takes the filter and
    equals the oracle's top-k of the fp32 score matrix."""
    import scipy.sparse as sp
    import tensorrec_amd as T
    rng = np.random.default_rng(9)
    n_u, n_i, d, k = 257, 20000, 64, 10
    uf = sp.identity(n_u, dtype=np.float32, format="csr")
    itf = sp.identity(n_i, dtype=np.float32, format="csr")
    inter = sp.csr_matrix((np.ones(n_u, np.float32), (np.arange(n_u), rng.integers(0, n_i, n_u))), shape=(n_u, n_i))
    model = T.TensorRec(n_components=d, seed=1)
    model.fit(inter, uf, itf, epochs=1)
    w = model.get_weights()
    uu = O.spmm_exact(uf, w["linear_weights_user_0"])
    vv = O.spmm_exact(itf, w["linear_weights_item"])
    ub = O.spmm_exact(uf, w["user_feature_biases"]).reshape(-1)
    ib = O.spmm_exact(itf, w["item_feature_biases"]).reshape(-1)
    ops.LAST_FILTER_STATS.clear()
    vals, idx = model.predict_top_k(uf, itf, k=k)
    assert ops.LAST_FILTER_STATS.get("users") == n_u            # the filtered path ran
    rv, ri = exact_reference(uu, vv, k, ub, ib)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def test_k1_filter_epilogue_equals_gather_plus_prep(ops):
    """trec_spmm_csr_filter (K1 emitting the filter's operand) == trec_spmm_csr followed by trec_score_prep_filter: the fp32
    rows and their bf16 images bit for bit, the norms to fp32 summation order, and the same exact top-k."""
    import scipy.sparse as sp
    from tensorrec_amd.sparse import SparseFeatures
    rng = np.random.default_rng(12)
    n_u, n_i, F, d, k = 400, 30000, 5000, 128, 10
    xu = sp.random(n_u, F, density=0.004, random_state=1, dtype=np.float32, format="csr")
    xi = sp.random(n_i, F, density=0.002, random_state=2, dtype=np.float32, format="csr")
    w_u = dev(rng.standard_normal((F, d)).astype(np.float32))
    w_i = dev(rng.standard_normal((F, d)).astype(np.float32))
    ib = dev(rng.standard_normal(n_i).astype(np.float32))
    fu, fi = SparseFeatures(xu, "cuda"), SparseFeatures(xi, "cuda")
    a_u = ops.spmm_filter_operand(fu, w_u)
    a_i = ops.spmm_filter_operand(fi, w_i, bias=ib, want_gstats=True)
    r_u = ops.spmm_raw(fu.indptr, fu.indices, fu.values, None, n_u, fu.nnz, w_u)
    r_i = ops.spmm_raw(fi.indptr, fi.indices, fi.values, None, n_i, fi.nnz, w_i)
    b_u = ops.score_prep_filter(r_u)
    b_i = ops.score_prep_filter(r_i, bias=ib, want_gstats=True)
    for a, b in ((a_u, b_u), (a_i, b_i)):
        assert torch.equal(a.f32, b.f32) and torch.equal(a.bf16, b.bf16)
        assert torch.allclose(a.stats, b.stats, rtol=1e-5, atol=1e-30)
    assert torch.allclose(a_i.gstats, b_i.gstats, rtol=1e-5)
    va, ia = ops.score_topk_filtered(a_u, a_i, k, None, ib)
    vb, ib_ = ops.score_topk_filtered(b_u, b_i, k, None, ib)
    assert torch.equal(va, vb) and torch.equal(ia, ib_)


def test_wide_second_pass_certifies_users_beyond_the_first_pass_capacities(ops):
    """A few items with 12x the typical norm inflate the filter's bound (it uses the catalogue-wide maxima): most users keep
    more than FILTER_KSEL = 48 superblocks and have more than 64 survivors -- what fitted models look like early in training.
    The wide second pass (320 slots, 16-entry lists, a finish without survivor limit) certifies them on the table that already
    exists; nobody reaches the fp32 MFMA fall-back, and the result is the oracle's."""
    rng = np.random.default_rng(31)
    n_u, n_i, d, k = 1500, 90_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[rng.permutation(n_i)[:25]] *= 12.0
    v *= rng.uniform(0.97, 1.0, (n_i, 1)).astype(np.float32)
    ub = (0.05 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.05 * rng.standard_normal(n_i)).astype(np.float32)
    du, dv, dub, dib = dev(u), dev(v), dev(ub), dev(ib)
    uop = ops.score_prep_filter(du)
    iop = ops.score_prep_filter(dv, bias=dib, want_gstats=True)
    vals, idx = ops.score_topk_filtered(uop, iop, k, dub, dib)
    stats = dict(ops.LAST_FILTER_STATS)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    assert stats["flagged_users"] > n_u // 4, stats                  # the first pass gave up on many users ...
    assert stats["flagged_after_wide_pass"] == 0, stats              # ... the wide pass certified all of them


# ---- Euclidean scores through the dot-product cascade (csrc/euclid_topk.hip) ----------------------------------------------
@pytest.mark.parametrize("d,n_u,n_i,k,bias_scale", [(128, 900, 300_000, 10, 0.0), (128, 700, 280_000, 10, 0.001), (64, 400, 40_000, 5, 0.0005),
                                                    (100, 300, 30_000, 12, 0.0), (128, 700, 280_000, 10, 0.02), (128, 500, 270_000, 10, 2.0),
                                                    (128, 600, 300_000, 20, 0.001), (128, 500, 290_000, 40, 0.0), (64, 400, 280_000, 48, 0.02),
                                                    (128, 300, 270_000, 30, 2.0)])
def test_euclidean_topk_through_the_dot_cascade_is_the_oracles(ops, d, n_u, n_i, k, bias_scale):
    """-sqrt(max(r_u - 2 u.i + r_i, 1e-16)) (+ biases): per user the nearest items are the largest u.i - r_i / 2, so the dot
    cascade lists the 16 nearest (k <= 12; the 32 / 64 nearest through the wide cascade up to k = 48), the reference chain re-scores
    them and a certificate decides per user; without it (item biases of 2.0: they outweigh the distance gaps) the exact fp32 path
    answers.  Values and ids == the oracle's, always."""
    rng = np.random.default_rng(d + n_u + k)
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = (rng.standard_normal((n_i, d)) * rng.uniform(0.7, 1.3, (n_i, 1))).astype(np.float32)
    ub = (bias_scale * rng.standard_normal(n_u)).astype(np.float32) if bias_scale else None
    ib = (bias_scale * rng.standard_normal(n_i)).astype(np.float32) if bias_scale else None
    du, dv = dev(u), dev(v)
    dub = dev(ub) if ub is not None else None
    dib = dev(ib) if ib is not None else None
    vals, idx = ops.score_topk_euclid_filtered(du, dv, k, dub, dib)
    stats = dict(ops.LAST_FILTER_STATS)
    _, u_sq, _ = ops.score_prep(du, ops.DTYPE_F32, want_sqnorm=True)
    _, v_sq, _ = ops.score_prep(dv, ops.DTYPE_F32, want_sqnorm=True)
    ref = O.score_dense_euclid_exact(u, v, u_sq.cpu().numpy(), v_sq.cpu().numpy(), ub, ib)
    rv, ri = O.topk_rows(ref, k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    # the cascade orders by u.i - r_i / 2 + lambda b_i (round 6): item biases of the size of the distance gaps between neighbours
    # (0.02 at distances of ~16: with the plain nearest-16 ordering 456 of 700 users had to be re-done) no longer break the
    # certificate; biases that dwarf every distance difference (2.0) are certified where the bound's curvature term allows
    if bias_scale <= 0.02:
        assert stats["euclid_uncertified_users"] <= n_u // 10, stats
    if k > 12:
        assert "%d nearest" % ops.euclid_candidates_for(k) in stats["route"] and "cascade, k up to" not in stats["route"], stats
    if bias_scale == 0.02:
        from tensorrec_amd import _native as N
        N.set_tuning("euclid_bias_in_order", 0)                     # the plain g ordering: exact as well, through the fp32 path
        try:
            vals0, idx0 = ops.score_topk_euclid_filtered(du, dv, k, dub, dib)
            stats0 = dict(ops.LAST_FILTER_STATS)
        finally:
            N.set_tuning("euclid_bias_in_order", 1)
        assert np.array_equal(idx0.cpu().numpy(), ri) and np.array_equal(vals0.cpu().numpy(), rv)
        assert stats0["euclid_uncertified_users"] > 4 * max(1, stats["euclid_uncertified_users"]), (stats0, stats)
    if n_i >= 262_144 and d in (64, 128):
        assert str(stats.get("prefilter", "")).startswith("int8"), stats      # the int8 -> bf16 -> fp32 cascade itself ran


def test_euclidean_predict_top_k_through_the_public_api(ops):
    """TensorRec.predict_top_k for EuclideanSimilarityPredictionGraph takes the filtered route and equals predict_rank's order."""
    import scipy.sparse as sp
    import tensorrec_amd as T
    rng = np.random.RandomState(4)
    n_u, n_i, d = 200, 20_000, 32
    uf = sp.random(n_u, 40, density=0.2, random_state=rng, format="csr", dtype=np.float32)
    itf = sp.hstack([sp.identity(n_i, format="csr", dtype=np.float32),
                     sp.random(n_i, 6, density=0.3, random_state=rng, format="csr", dtype=np.float32)], format="csr")
    model = T.TensorRec(n_components=d, seed=2, prediction_graph=T.prediction_graphs.EuclideanSimilarityPredictionGraph())
    model.build(uf.shape[1], itf.shape[1])
    w = model.get_weights()
    w["item_feature_biases"] = (0.01 * rng.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
    w["user_feature_biases"] = (0.3 * rng.standard_normal(w["user_feature_biases"].shape)).astype(np.float32)
    model.set_weights(w)
    vals, idx = model.predict_top_k(uf, itf, k=10)
    assert "euclidean" in str(ops.LAST_FILTER_STATS.get("route", ""))
    scores = model.predict(uf, itf)
    rv, ri = O.topk_rows(scores, 10)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


# ---- the all-superblocks tier of the wide pass on a catalogue above 4096 superblocks ---------------------------
def test_wide_pass_tier_two_respects_the_collect_limit_above_two_million_items(ops):
    """2.2M items = 4,297 superblocks: tier 2 used to ask trec_topk_collect_blocks for more slots than it has and the call
    raised.  Users built to keep (nearly) every superblock are flagged through both tiers and must come back exact from the
    fp32 path; the others take the filter."""
    rng = np.random.default_rng(77)
    n_u, n_i, d, k = 96, 2_200_000, 32, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u[:2] *= 1e-4                              # two users whose scores all lie within the bound of each other: every superblock is kept
    du, dv = dev(u), dev(v)
    uop = ops.score_prep_filter(du)
    iop = ops.score_prep_filter(dv, want_gstats=True)
    v2 = dv.clone()
    v2[::400_000] *= 30.0                      # a few huge items inflate the item-side maxima: the bound is loose for everybody
    iop2 = ops.score_prep_filter(v2, want_gstats=True)
    vals, idx = ops.score_topk_filtered(uop, iop2, k)
    stats = dict(ops.LAST_FILTER_STATS)
    ref = O.score_dense_exact(u, v2.cpu().numpy())
    rv, ri = O.topk_rows(ref, k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    assert stats["flagged_users"] >= 1, stats


# ---- 17 <= k <= 64 through the cascade (ops.score_topk_filtered_wide) ------------------------------------------------------------
@pytest.mark.parametrize("d,n_u,n_i,k,biased", [(128, 700, 300_000, 17, True), (128, 500, 280_000, 64, True), (64, 400, 270_000, 32, False),
                                                (128, 300, 40_000, 40, True), (128, 260, 1_000_000, 64, True)])
def test_wide_k_topk_through_the_cascade_is_the_oracles(ops, d, n_u, n_i, k, biased):
    """k above the 16 slots of the fused lists: the cascade's int8 / bf16 stages with tau = the k-th largest of the chunks'
    lower-bound lists, 1,024 candidate slots per user, the reference chain on every listed pair and the k best of each list.
    Values and ids == the oracle's (tf.nn.top_k of recommendation_graphs.py:80 over prediction_graphs.py:49-50)."""
    rng = np.random.default_rng(d + n_u + k)
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = (rng.standard_normal((n_i, d)) * rng.uniform(0.8, 1.2, (n_i, 1))).astype(np.float32)
    ub = (0.1 * rng.standard_normal(n_u)).astype(np.float32) if biased else None
    ib = (0.1 * rng.standard_normal(n_i)).astype(np.float32) if biased else None
    du, dv = dev(u), dev(v)
    dub = dev(ub) if biased else None
    dib = dev(ib) if biased else None
    uop = ops.score_prep_filter(du, sort_users=True, k=k, user_bias=dub)
    iop = ops.score_prep_filter(dv, bias=dib, want_gstats=True)
    vals, idx = ops.score_topk_filtered_wide(uop, iop, k, dub, dib, item_index_base=500)
    stats = dict(ops.LAST_FILTER_STATS)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx.cpu().numpy(), ri + 500) and np.array_equal(vals.cpu().numpy(), rv), stats
    # (k = 64 of 547 superblocks: the 64th largest lower bound keeps more than 45 % of all pairs -- "too loose", score slabs answer;
    # from ~1,000 superblocks on the lists do)
    if n_i >= 262_144 and (k <= 32 or n_i >= 1_000_000):
        assert stats["prefilter"] == "int8" and stats["flagged_users"] <= n_u // 10, stats       # the lists answered, not the fallback
        assert stats["prerefined_pairs"] == n_u * k, stats                     # ... behind the pre-refinement of each user's k best superblocks
        # the same call without it (tuning wide_prerefine = 0): the same bits from longer lists
        from tensorrec_amd import _native as N
        N.set_tuning("wide_prerefine", 0)
        try:
            uop = ops.score_prep_filter(du, sort_users=True, k=k, user_bias=dub)
            iop = ops.score_prep_filter(dv, bias=dib, want_gstats=True)
            vals0, idx0 = ops.score_topk_filtered_wide(uop, iop, k, dub, dib, item_index_base=500)
            stats0 = dict(ops.LAST_FILTER_STATS)
        finally:
            N.set_tuning("wide_prerefine", 1)
        assert torch.equal(vals0, vals) and torch.equal(idx0, idx)
        assert stats0["prerefined_pairs"] == 0 and stats0["refined_rows"] >= stats["refined_rows"], (stats0, stats)


def test_wide_k_predict_top_k_through_the_public_api(ops):
    """TensorRec.predict_top_k(k=40) on a catalogue the cascade runs on takes the wide route and equals the dense prediction's order."""
    import scipy.sparse as sp
    import tensorrec_amd as T
    rng = np.random.RandomState(6)
    n_u, n_i, d = 180, 270_000, 64
    uf = sp.random(n_u, 30, density=0.3, random_state=rng, format="csr", dtype=np.float32)
    itf = sp.hstack([sp.identity(n_i, format="csr", dtype=np.float32),
                     sp.random(n_i, 4, density=0.3, random_state=rng, format="csr", dtype=np.float32)], format="csr")
    model = T.TensorRec(n_components=d, seed=3)
    model.build(uf.shape[1], itf.shape[1])
    w = model.get_weights()
    w["item_feature_biases"] = (0.05 * rng.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
    model.set_weights(w)
    vals, idx = model.predict_top_k(uf, itf, k=40)
    assert "cascade, k up to" in str(ops.LAST_FILTER_STATS.get("route", "")), dict(ops.LAST_FILTER_STATS)
    scores = model.predict(uf, itf)
    rv, ri = O.topk_rows(scores, 40)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


@pytest.mark.parametrize("n_i,k", [(300_000, 40), (70_001, 17), (9_000, 3)])
def test_topk_from_scores_block_maxima_bound_selects_the_same_entries(ops, n_i, k):
    """Long rows of continuous scores: the k-th largest of the 512-entry block maxima selects a superset of the row's k best (two
    streaming passes instead of a row-wise selection); the result equals the oracle's order and the form with the exact k-th value
    (tuning topk_slab_block_bound = 0) bit for bit -- also with a ragged last block, +inf entries and a few duplicated values."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(n_i + k)
    s = rng.standard_normal((37, n_i)).astype(np.float32)
    s[2, 5:60:7] = np.inf
    s[4, n_i - 3:] = 9.0                                                         # the row's best entries in the ragged tail, tied
    s[7, 100:100 + 2 * k] = s[7].max()                                          # 2k copies of the maximum: ties across the k-th place
    rv, ri = O.topk_rows(s, k)
    ds = dev(s)
    vals, idx = ops.topk_from_scores(ds, k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    N.set_tuning("topk_slab_block_bound", 0)
    try:
        vals0, idx0 = ops.topk_from_scores(ds, k)
    finally:
        N.set_tuning("topk_slab_block_bound", 1)
    assert torch.equal(vals0, vals) and torch.equal(idx0, idx)


@pytest.mark.parametrize("n_i,k", [(20_000, 40), (9_000, 3), (300, 5)])
def test_topk_from_scores_selects_by_the_kth_value_and_keeps_the_tie_rule(ops, n_i, k):
    """ops.topk_from_scores on long rows: the k-th largest value selects, trec_topk_merge orders -- equal to the oracle's
    tf.nn.top_k order (value desc, index asc; recommendation_graphs.py:80) on rows with heavy ties, +inf entries, and -- through the
    exact-rank form it falls back to -- rows whose k-th value is -inf or that hold NaN-free but mostly -inf entries."""
    rng = np.random.default_rng(n_i + k)
    s = np.round(rng.standard_normal((23, n_i)) * 4).astype(np.float32) / 4      # quarter-integer values: ties everywhere
    s[3, 2:40:9] = np.inf
    s[5, 10:] = -np.inf                                                          # fewer finite entries than k (when k > 10)
    s[6, :] = 1.5                                                                # one value: the first k indices
    rv, ri = O.topk_rows(s, k)
    vals, idx = ops.topk_from_scores(dev(s), k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    # without the degenerate rows the threshold form itself answers
    s2 = np.delete(s, [5, 6], axis=0)
    rv, ri = O.topk_rows(s2, k)
    vals, idx = ops.topk_from_scores(dev(s2), k)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
