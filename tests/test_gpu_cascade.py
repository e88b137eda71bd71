"""K2q / K2c -- the int8 -> bf16 -> fp32 cascade of the exact top-k (csrc/score_blockmax_i8.hip, csrc/topk_cascade.hip,
ops.score_topk_filtered(prefilter="int8")).

Bar: values AND item ids bit-identical to the oracle's fp32 restatement of tf.matmul + tf.nn.top_k
(tensorrec/prediction_graphs.py:49-50, tensorrec/recommendation_graphs.py:80) -- the int8 stage only decides which
(superblock, user) pairs the bf16 stage looks at; the int8 maxima are exact integer arithmetic (checked against an integer
reference), the bound e(u, s) must dominate every observed |int8 score - fp32 score|, and the row-wise compaction must
list exactly the pairs at or above the floor."""
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


def run_cascade(ops, u, v, k, ub=None, ib=None, normalize=False, sort_users=True, **kw):
    """sort_users=True is the product path (predict_top_k): users sorted by int8 scale class, one scale per int8 workgroup;
    False: ONE scale for all users (round 2's form, kept as the A/B reference)."""
    du, dv = dev(u), dev(v)
    dub = dev(ub) if ub is not None else None
    dib = dev(ib) if ib is not None else None
    uop = ops.score_prep_filter(du, normalize=normalize, sort_users=sort_users, k=k)
    iop = ops.score_prep_filter(dv, normalize=normalize, bias=dib, want_gstats=True)
    vals, idx = ops.score_topk_filtered(uop, iop, k, dub, dib, prefilter="int8", **kw)
    return vals.cpu().numpy(), idx.cpu().numpy(), dict(ops.LAST_FILTER_STATS), uop, iop


@pytest.mark.parametrize("d,biased,n_u,n_i", [(128, True, 700, 40000 + 77), (128, False, 1500, 66000), (64, True, 513, 30011),
                                              (100, True, 300, 25000), (32, False, 1024, 20480), (128, True, 3, 5200)])
def test_cascade_topk_bit_exact_vs_oracle(ops, d, biased, n_u, n_i):
    rng = np.random.default_rng(d + biased + n_u)
    k = 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ub = (0.1 * rng.standard_normal(n_u)).astype(np.float32) if biased else None
    ib = (0.1 * rng.standard_normal(n_i)).astype(np.float32) if biased else None
    vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri)
    assert np.array_equal(vals, rv)
    # (small catalogues: few superblocks, the k-th largest maximum is not selective and the cascade may hand stage 1 back to
    # bf16 -- "int8 (too loose ...)"; either way the result above is exact)
    # d = 32 (kpad 32) is outside the int8 kernel's shapes: plain bf16 filter, no "prefilter" entry
    assert stats.get("prefilter", "int8" if d == 32 else "").startswith("int8")
    assert stats["flagged_users"] <= max(1, n_u // 20)


def test_cascade_heterogeneous_rows_and_large_biases(ops):
    """Row norms spread over three decades, biases as large as the scores, a few huge outlier items: the int8 bound
    is loose here (the global item scale follows the outliers) -- the result must stay exact whichever stage 1 ran."""
    rng = np.random.default_rng(11)
    n_u, n_i, d, k = 600, 50000, 128, 10
    u = (rng.standard_normal((n_u, d)) * rng.lognormal(0, 1.5, (n_u, 1))).astype(np.float32)
    v = (rng.standard_normal((n_i, d)) * rng.lognormal(0, 1.5, (n_i, 1))).astype(np.float32)
    v[::5000] *= 50
    ub = rng.standard_normal(n_u).astype(np.float32) * 5
    ib = rng.standard_normal(n_i).astype(np.float32) * 5
    vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def test_cascade_user_batches_share_the_item_operand(ops):
    """predict_top_k walks the users in batches: the item rows are quantised once, every batch brings its own user scale
    (the scale product and the integer item biases are re-derived) -- each batch must be exact."""
    rng = np.random.default_rng(13)
    n_u, n_i, d, k = 900, 300_000, 64, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    u[300:600] *= 7.0                                                        # the second batch has a different scale
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = rng.standard_normal(n_u).astype(np.float32)
    ib = rng.standard_normal(n_i).astype(np.float32)
    dib = dev(ib)
    iop = ops.score_prep_filter(dev(v), bias=dib, want_gstats=True)
    used = []
    for s in range(0, n_u, 300):
        uop = ops.score_prep_filter(dev(u[s:s + 300]), sort_users=(s != 300))   # both forms of the user side against one item operand
        vals, idx = ops.score_topk_filtered(uop, iop, k, dev(ub[s:s + 300]), dib, prefilter="int8")
        used.append(ops.LAST_FILTER_STATS["prefilter"])
        rv, ri = O.topk_rows(O.score_dense_exact(u[s:s + 300], v, ub[s:s + 300], ib), k)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    assert "int8" in used                                                    # 586 superblocks: the cascade itself ran


@pytest.mark.parametrize("k,d", [(16, 64), (12, 128), (1, 128)])
def test_cascade_other_k_on_a_selective_catalogue(ops, k, d):
    """k = 11..16 takes the 16-slot form of the int8 kernel's register lists, k = 1 the 10-slot one with a single live
    slot; 300,000 items = 586 superblocks: the cascade itself runs (no fall-back)."""
    rng = np.random.default_rng(100 + k)
    n_u, n_i = 260, 300_000
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = rng.standard_normal(n_u).astype(np.float32) * 0.5
    ib = rng.standard_normal(n_i).astype(np.float32) * 0.5
    vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["prefilter"] == "int8" and stats["flagged_users"] <= 13


def test_cascade_remembers_a_catalogue_it_is_too_loose_for(ops):
    """A small catalogue (16 superblocks, k = 10) is not selective at all: the first call tries the int8 stage, falls back, and
    marks the item operand; the next user batch goes straight to the bf16 filter.  Both exact."""
    rng = np.random.default_rng(77)
    n_u, n_i, d, k = 3000, 8192, 128, 10                     # (few users never overflow: the list has room for padding)
    u = rng.standard_normal((2 * n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    iop = ops.score_prep_filter(dev(v), want_gstats=True)
    seen = []
    for b in range(2):
        uop = ops.score_prep_filter(dev(u[b * n_u:(b + 1) * n_u]))
        vals, idx = ops.score_topk_filtered(uop, iop, k, prefilter="int8")
        seen.append(ops.LAST_FILTER_STATS.get("prefilter"))
        rv, ri = O.topk_rows(O.score_dense_exact(u[b * n_u:(b + 1) * n_u], v), k)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    assert seen[0] == "int8 (too loose: bf16 stage 1 instead)" and seen[1] is None and iop.cascade_too_loose


def test_cascade_ties_and_integer_data(ops):
    """Small-integer operands: int8 quantisation is exact only by luck of the scale, every score ties with many others;
    ids must follow tf.nn.top_k's lower-index-first order."""
    rng = np.random.default_rng(3)
    n_u, n_i, d, k = 400, 30000, 64, 10
    u = rng.integers(-2, 3, (n_u, d)).astype(np.float32)
    v = rng.integers(-2, 3, (n_i, d)).astype(np.float32)
    vals, idx, stats, _, _ = run_cascade(ops, u, v, k)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def test_cascade_non_finite_rows_fall_back(ops):
    rng = np.random.default_rng(4)
    n_u, n_i, d, k = 300, 20000, 128, 5
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    u[7, 3] = np.inf
    v[123, 5] = np.nan
    vals, idx, stats, _, iop = run_cascade(ops, u, v, k)
    uref = ops.score_prep_filter(dev(u))                                     # (the cascade's own user operand is sorted by class)
    ev, ei = ops.score_topk(uref.f32, iop.f32, ops.DTYPE_F32, uref.kpad, k, method="two_stage")
    assert np.array_equal(idx, ei.cpu().numpy())
    assert np.array_equal(vals, ev.cpu().numpy(), equal_nan=True)


def _pair_err(nx, ex, cu, yh, dy, db, kdim):
    """i8_pair_err (csrc/score_common.hpp) in float32, operation for operation."""
    f = np.float32
    ck = f(kdim + 6) * f(2.98023224e-07)
    e = nx * (dy + ck * yh) + ex * yh + db + cu
    return e * f(1.001953125) + f(1e-30)


def test_int8_table_is_exact_integer_arithmetic_and_bound_holds(ops):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(8)
    n_u, n_i, d, sb = 640, 30000 + 5, 128, 512
    u = rng.standard_normal((n_u, d)).astype(np.float32) * rng.uniform(0.5, 2, (n_u, 1)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32) * rng.uniform(0.3, 3, (n_i, 1)).astype(np.float32)
    ub = rng.standard_normal(n_u).astype(np.float32) * 0.3
    ib = rng.standard_normal(n_i).astype(np.float32) * 0.3
    uop = ops.score_prep_filter(dev(u))
    iop = ops.score_prep_filter(dev(v), bias=dev(ib), want_gstats=True)
    ops.score_prep_i8_pair(uop, iop, dev(ib), sb)
    n_sb = (n_i + sb - 1) // sb
    a = iop.scales.cpu().numpy()[0]
    sbs = iop.sb_stats.cpu().numpy()
    b_row = np.repeat(sbs[:, 0], sb)[:n_i]                                    # one item scale per superblock
    uq, iq = uop.i8.cpu().numpy().astype(np.int64), iop.i8.cpu().numpy().astype(np.int64)
    # quantisation as documented, error norms as measured
    for s in range(n_sb):
        blk = v[s * sb:(s + 1) * sb]
        assert sbs[s, 0] == np.abs(blk).max() / np.float32(127)
    assert np.array_equal(iq, np.clip(np.rint(v * (np.float32(1) / b_row)[:, None]), -127, 127).astype(np.int64))
    assert all(np.abs(iq[s * sb:(s + 1) * sb]).max() == 127 for s in range(n_sb))          # every superblock uses its full range
    dy = np.linalg.norm(v - iq * b_row[:, None], axis=1)
    np.testing.assert_allclose(iop.stats8.cpu().numpy()[:, 1], dy, rtol=1e-4)
    np.testing.assert_allclose(uop.stats8.cpu().numpy()[:, 1], np.linalg.norm(u - uq * a, axis=1), rtol=1e-4)
    st8 = iop.stats8.cpu().numpy()
    for s in (0, 7, n_sb - 1):
        rows = slice(s * sb, min(n_i, (s + 1) * sb))
        assert sbs[s, 1] == (st8[rows, 0] + st8[rows, 1]).max() and sbs[s, 2] == st8[rows, 1].max()
    sp_row = np.float32(a) * b_row
    bq = iop.bias_q.cpu().numpy().astype(np.int64)
    assert np.array_equal(bq, np.rint(ib / sp_row).astype(np.int64))
    # the table: exact integer arithmetic, converted once per (user, superblock); the chunks' lists: the 10 largest lower bounds
    dub = dev(ub)
    uerr = torch.empty((n_u, 4), dtype=torch.float32, device="cuda")
    N.call("trec_score_user_err_i8", N.ptr(uop.stats8), N.ptr(dub), N.ptr(iop.gstats8), d, n_u, N.ptr(iop.scales), None, 0,
           N.ptr(uerr))
    assert np.all(uerr.cpu().numpy()[:, 3] == a)                             # one class: every user carries the one scale
    n_chunks = 3
    chunk_len, n_ch = ops.blockmax_i8_chunks(n_i, n_chunks, sb)
    table = torch.empty((n_sb, n_u), dtype=torch.float32, device="cuda")
    ctop = torch.empty((n_ch * 10, n_u), dtype=torch.float32, device="cuda")
    N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), d, n_u, n_i, N.ptr(dub), N.ptr(iop.bias_q),
           N.ptr(iop.scales), N.ptr(iop.sb_stats), sb, n_chunks, N.ptr(table), n_u, N.ptr(uerr), N.ptr(ctop), 10, None, None, 0)
    s_int = uq @ iq.T + bq[None, :]
    pad = n_sb * sb - n_i
    m_int = np.concatenate([s_int, np.full((n_u, pad), np.iinfo(np.int64).min)], 1).reshape(n_u, n_sb, sb).max(2)
    sp_sb = np.float32(a) * sbs[:, 0]
    want = (m_int.astype(np.float32) * sp_sb[None, :] + ub[:, None]).T
    got = table.cpu().numpy()
    assert np.array_equal(got, want)
    ue = uerr.cpu().numpy()
    # sb_stats[s][3] is the bias quantisation error PER UNIT of user scale: the bound takes a_u times it
    e = _pair_err(ue[:, 0][None, :], ue[:, 1][None, :], ue[:, 2][None, :], sbs[:, 1][:, None], sbs[:, 2][:, None],
                  ue[:, 3][None, :] * sbs[:, 3][:, None], d)                  # [n_sb, n_u]
    lb = got - e
    spc = chunk_len // sb
    ct = ctop.cpu().numpy().reshape(n_ch, 10, n_u)
    for c in range(n_ch):
        ref = -np.sort(-lb[c * spc:(c + 1) * spc], axis=0)[:10]
        assert np.array_equal(ct[c][:ref.shape[0]], ref) and np.all(ct[c][ref.shape[0]:] == -np.inf)
    # the TAGGED lists (top_k | 0x100, the pre-refinement's input): every entry is a lower bound at most 2^-11 of its size below the
    # untagged one, and its low 12 bits name the superblock (inside the chunk) it came from; the table does not change
    table_t = torch.empty((n_sb, n_u), dtype=torch.float32, device="cuda")
    ctop_t = torch.empty((n_ch * 10, n_u), dtype=torch.float32, device="cuda")
    N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), d, n_u, n_i, N.ptr(dub), N.ptr(iop.bias_q),
           N.ptr(iop.scales), N.ptr(iop.sb_stats), sb, n_chunks, N.ptr(table_t), n_u, N.ptr(uerr), N.ptr(ctop_t), 10 | 0x100, None, None, 0)
    assert np.array_equal(table_t.cpu().numpy(), got)
    tg = ctop_t.cpu().numpy().reshape(n_ch, 10, n_u)
    bits = tg.view(np.int32).astype(np.int64)
    key = np.where(bits >= 0, bits, -(bits & 0x7FFFFFFF))                     # the monotone integer key of lb_tag (score_common.hpp)
    tag = key & 0xFFF
    for c in range(n_ch):
        n_here = min(10, lb[c * spc:(c + 1) * spc].shape[0])
        t, r = tg[c][:n_here], ct[c][:n_here]
        assert np.all(t <= r) and np.all(r - t <= np.abs(r) * 2.0 ** -10 + 1e-30)
        assert np.all(tg[c][n_here:] == -np.inf)
        # the tagged superblock's own (untagged) lower bound stands right above the entry; lists sorted by tagged value may order two
        # bounds within 2^-11 of each other differently, so the comparison is per entry, and the smallest source is (nearly) the 10th
        src = lb[c * spc + tag[c][:n_here], np.arange(n_u)[None, :]]
        assert np.all(t <= src) and np.all(src - t <= np.abs(src) * 2.0 ** -10 + 1e-30), c
        assert np.all(src.min(axis=0) >= r[-1] - np.abs(r[-1]) * 2.0 ** -10 - 1e-30), c
        assert all(len(np.unique(tag[c][:n_here, j])) == n_here for j in range(0, n_u, 37))     # distinct superblocks per user
    # the bound: e(u, s) against the observed |int8 score - fp32 score| of every item of the superblock
    s8 = s_int.astype(np.float64) * sp_row[None, :].astype(np.float64) + ub[:, None]
    s32 = O.score_dense_exact(u, v, ub, ib).astype(np.float64)
    diff = np.concatenate([np.abs(s8 - s32), np.zeros((n_u, pad))], 1).reshape(n_u, n_sb, sb).max(2).T    # [n_sb, n_u]
    assert np.all(diff <= e)
    assert np.median(e / np.maximum(diff, 1e-12)) < 40                        # ... and is not absurdly loose


def test_int8_bound_is_tight_but_holds_on_aligned_quantisation_errors(ops):
    """Cauchy-Schwarz is tight when the rounding errors are parallel to the other operand.  Users x = a0 (q + 0.49 s) and
    items y = b0 (r + 0.49 t) with power-of-two scales planted through the rows' largest elements (so the prep kernels
    derive exactly a0 / b0) quantise to q / r with error 0.49 a0 s / 0.49 b0 t; s and t follow the signs of chosen partner
    rows, so the two error terms add up coherently.  The bound must still dominate, with little room to spare."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(21)
    n_u, n_i, d, sb = 256, 2048, 128, 512
    a0, b0 = np.float32(2.0 ** -8), np.float32(2.0 ** -9)
    q = rng.integers(-100, 101, (n_u, d)).astype(np.float32)
    r = rng.integers(-100, 101, (n_i, d)).astype(np.float32)
    r[r == 0] = 1
    q[q == 0] = 1
    partner = rng.integers(0, n_i, n_u)                                      # user u's errors follow item partner[u] ...
    s_ = np.sign(r[partner])
    t_ = np.sign(q[rng.integers(0, n_u, n_i)])                               # ... item i's errors some user's row
    t_[partner] = np.sign(q)                                                 # (the partner's own errors follow ITS user: both terms add)
    x = a0 * (q + np.float32(0.49) * s_)
    y = b0 * (r + np.float32(0.49) * t_)
    x[:, 0] = 127 * a0                                                       # max |x| = 127 a0 < 4 rms: the user scale is a0
    for s in range(n_i // sb):
        y[s * sb, 1] = 127 * b0                                              # every superblock's scale is b0
    uop = ops.score_prep_filter(dev(x))
    iop = ops.score_prep_filter(dev(y), want_gstats=True)
    ops.score_prep_i8_pair(uop, iop, None, sb)
    assert iop.scales.cpu().numpy()[0] == a0 and np.all(iop.sb_stats.cpu().numpy()[:, 0] == b0)
    uq, iq = uop.i8.cpu().numpy().astype(np.int64), iop.i8.cpu().numpy().astype(np.int64)
    s8 = (uq @ iq.T).astype(np.float64) * float(a0) * float(b0)
    s32 = O.score_dense_exact(x, y).astype(np.float64)
    uerr = torch.empty((n_u, 4), dtype=torch.float32, device="cuda")
    N.call("trec_score_user_err_i8", N.ptr(uop.stats8), None, N.ptr(iop.gstats8), d, n_u, N.ptr(iop.scales), None, 0, N.ptr(uerr))
    ue, sbs = uerr.cpu().numpy(), iop.sb_stats.cpu().numpy()
    e = _pair_err(ue[:, 0][None, :], ue[:, 1][None, :], ue[:, 2][None, :], sbs[:, 1][:, None], sbs[:, 2][:, None],
                  ue[:, 3][None, :] * sbs[:, 3][:, None], d)                  # [n_sb, n_u]
    diff = np.abs(s8 - s32).reshape(n_u, n_i // sb, sb).max(2).T
    assert np.all(diff <= e)
    at_partner = np.abs(s8 - s32)[np.arange(n_u), partner] / e[partner // sb, np.arange(n_u)]
    assert np.median(at_partner) > 0.6                                        # coherent errors use most of the bound


def test_rows_compaction_lists_exactly_the_pairs_whose_upper_bound_reaches_the_threshold(ops):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(2)
    for n_sb, n_u, stride in ((37, 5000, 5000), (9, 1023, 1024), (130, 2050, 2051)):
        table = rng.standard_normal((n_sb, stride)).astype(np.float32)
        thr = rng.uniform(0.5, 2.5, n_u).astype(np.float32)
        thr[5] = -np.inf                                                       # a user that keeps every superblock
        thr[6] = np.inf
        uerr = rng.uniform(0.0, 0.3, (n_u, 4)).astype(np.float32)              # {||x||, ||x - a q||, cu, a}
        sbs = rng.uniform(0.0, 1.0, (n_sb, 4)).astype(np.float32)
        sbs[3, 2] = np.inf                                                     # a superblock with an unusable bound: always kept
        dt, dth, due, dsb = dev(table), dev(thr), dev(uerr), dev(sbs)
        n_ublk = N.query("trec_topk_rows_user_blocks", n_u)
        block_off = torch.empty((n_sb * n_ublk,), dtype=torch.int32, device="cuda")
        row_total = torch.empty((n_sb,), dtype=torch.int32, device="cuda")
        row_pad = torch.empty((n_sb,), dtype=torch.int32, device="cuda")
        pstart = torch.empty((n_sb + 1,), dtype=torch.int64, device="cuda")
        with np.errstate(invalid="ignore", over="ignore"):
            # tile_bits (csrc/topk_cascade.hip), operation for operation in float32 (fma through float64: exact product)
            f32, f64 = np.float32, np.float64
            infl, ck = f32(1.0029296875), f32(128 + 6) * f32(2.98023224e-07)
            A = ((sbs[:, 2] + ck * sbs[:, 1]) * infl)[:, None]
            B = (sbs[:, 1] * infl)[:, None]
            C = (sbs[:, 3] * infl)[:, None]
            cu = uerr[:, 2] * infl + f32(2e-30)

            def pred(x):
                y = np.nextafter(x, f32(-np.inf))
                y[x == -np.inf] = -np.inf
                return y
            f = pred(pred(pred(thr.copy())) - cu)
            fma = lambda a, b, c: (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)
            tv = table[:, :n_u]
            lhs = fma(np.broadcast_to(uerr[:, 0][None, :], tv.shape), np.broadcast_to(A, tv.shape),
                      fma(np.broadcast_to(uerr[:, 1][None, :], tv.shape), np.broadcast_to(B, tv.shape),
                          fma(np.broadcast_to(uerr[:, 3][None, :], tv.shape), np.broadcast_to(C, tv.shape), tv)))
            keep = ~(lhs < f[None, :])
            keep[:, ~(thr < np.inf)] = False          # a threshold of +inf keeps NOTHING, whatever the table or the bound holds
        want_rows = int(((keep.sum(1) + 511) // 512 * 512).sum())
        for cap_rows in (want_rows + 1024, want_rows, want_rows - 512):          # roomy, exact fit, one workgroup short
            status = torch.full((2,), -5, dtype=torch.int64, device="cuda")
            N.call("trec_topk_rows_count", N.ptr(dt), n_sb, n_u, stride, N.ptr(dth), N.ptr(due), N.ptr(dsb), 128,
                   N.ptr(block_off), N.ptr(row_total), N.ptr(row_pad), N.ptr(pstart), cap_rows, N.ptr(status))
            assert np.array_equal(row_total.cpu().numpy(), keep.sum(1))
            ps = pstart.cpu().numpy()
            assert np.array_equal(np.diff(ps), (keep.sum(1) + 511) // 512 * 512)
            assert status.tolist() == [want_rows, int(want_rows > cap_rows)]
            row_user = torch.full((cap_rows,), -7, dtype=torch.int32, device="cuda")
            rblock_chunk = torch.full((cap_rows // 512,), -7, dtype=torch.int32, device="cuda")
            N.call("trec_topk_rows_fill", N.ptr(dt), n_sb, n_u, stride, N.ptr(dth), N.ptr(due), N.ptr(dsb), 128,
                   N.ptr(block_off), N.ptr(row_total), N.ptr(pstart), cap_rows, N.ptr(status), N.ptr(row_user),
                   N.ptr(rblock_chunk))
            ru, rc = row_user.cpu().numpy(), rblock_chunk.cpu().numpy()
            if want_rows > cap_rows:                                              # overflow: every workgroup idle, no row written
                assert np.all(rc == -1) and np.all(ru == -7)
                continue
            for s in range(n_sb):
                users = np.nonzero(keep[s])[0]
                seg = ru[ps[s]:ps[s + 1]]
                assert np.array_equal(seg[:len(users)], users) and np.all(seg[len(users):] == -1)
                assert np.all(rc[ps[s] // 512:ps[s + 1] // 512] == s)
            assert np.all(rc[want_rows // 512:] == -1)


def test_rows_collect_one_pass_keeps_the_same_pairs(ops):
    """trec_topk_rows_collect: per superblock the same SET of users as the two-pass compaction (the order follows the
    atomics), the counts, the overflow flag when a superblock keeps more than the capacity."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(12)
    for n_sb, n_u, stride, rcap in ((37, 5000, 5000, 2048), (130, 2050, 2051, 1024), (9, 3000, 3000, 512)):
        table = rng.standard_normal((n_sb, stride)).astype(np.float32)
        thr = rng.uniform(0.5, 2.5, n_u).astype(np.float32)
        thr[5] = -np.inf
        uerr = rng.uniform(0.0, 0.3, (n_u, 4)).astype(np.float32)
        sbs = rng.uniform(0.0, 1.0, (n_sb, 4)).astype(np.float32)
        with np.errstate(invalid="ignore", over="ignore"):
            f32, f64 = np.float32, np.float64
            infl, ck = f32(1.0029296875), f32(128 + 6) * f32(2.98023224e-07)
            A = ((sbs[:, 2] + ck * sbs[:, 1]) * infl)[:, None]
            B = (sbs[:, 1] * infl)[:, None]
            C = (sbs[:, 3] * infl)[:, None]
            cu = uerr[:, 2] * infl + f32(2e-30)

            def pred(x):
                y = np.nextafter(x, f32(-np.inf))
                y[x == -np.inf] = -np.inf
                return y
            f = pred(pred(pred(thr.copy())) - cu)
            fma = lambda a, b, c: (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)
            tv = table[:, :n_u]
            lhs = fma(np.broadcast_to(uerr[:, 0][None, :], tv.shape), np.broadcast_to(A, tv.shape),
                      fma(np.broadcast_to(uerr[:, 1][None, :], tv.shape), np.broadcast_to(B, tv.shape),
                          fma(np.broadcast_to(uerr[:, 3][None, :], tv.shape), np.broadcast_to(C, tv.shape), tv)))
            keep = ~(lhs < f[None, :])
        row_count = torch.zeros((n_sb,), dtype=torch.int32, device="cuda")
        row_user = torch.full((n_sb * rcap,), -7, dtype=torch.int32, device="cuda")
        status = torch.full((2,), -5, dtype=torch.int64, device="cuda")
        dt, dth, due, dsb = dev(table), dev(thr), dev(uerr), dev(sbs)           # (kept alive across the launch)
        N.call("trec_topk_rows_collect", N.ptr(dt), n_sb, n_u, stride, N.ptr(dth), N.ptr(due), N.ptr(dsb), 128, rcap,
               N.ptr(row_count), N.ptr(row_user), N.ptr(status))
        cnt = row_count.cpu().numpy()
        assert np.array_equal(cnt, keep.sum(1))
        ru = row_user.cpu().numpy().reshape(n_sb, rcap)
        over = bool((cnt > rcap).any())
        assert status.tolist() == [int(((np.minimum(cnt, rcap) + 511) // 512 * 512).sum()), int(over)]
        for s in range(n_sb):
            users = np.nonzero(keep[s])[0]
            got = ru[s, :min(cnt[s], rcap)]
            if cnt[s] <= rcap:
                assert np.array_equal(np.sort(got), users) and np.all(ru[s, cnt[s]:] == -7)
            else:
                assert len(np.unique(got)) == rcap and np.all(np.isin(got, users))


def test_grouped_bf16_blockmax_equals_the_full_kernel_on_the_listed_pairs(ops):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(6)
    n_u, n_i, d, sb = 1300, 20000 + 9, 128, 512
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub, ib = dev(rng.standard_normal(n_u).astype(np.float32)), dev(rng.standard_normal(n_i).astype(np.float32))
    uop = ops.score_prep_filter(dev(u))
    iop = ops.score_prep_filter(dev(v), bias=ib, want_gstats=True)
    n_sb = (n_i + sb - 1) // sb
    full = torch.empty((n_sb, n_u), dtype=torch.float32, device="cuda")
    N.call("trec_score_gemm_blockmax", N.ptr(uop.bf16), N.ptr(iop.bf16), ops.DTYPE_BF16, d, n_u, n_i, N.ptr(ub), N.ptr(ib),
           ops.MODE_DOT, None, None, sb, 2, N.ptr(full), n_u, 1 | 32)        # bit 5: the filters' 16x16x32 form, as the grouped launch
    keep = rng.random((n_sb, n_u)) < 0.3
    rows, chunks = [], []
    for s in range(n_sb):
        us = np.nonzero(keep[s])[0]
        padn = (-len(us)) % 512
        rows.append(np.concatenate([us, np.full(padn, -1)]))
        chunks += [s] * ((len(us) + padn) // 512)
    row_user = dev(np.concatenate(rows).astype(np.int32))
    rblock_chunk = dev(np.asarray(chunks, np.int32))
    table = torch.full((n_sb, n_u), -123.0, dtype=torch.float32, device="cuda")
    N.call("trec_score_gemm_blockmax_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), d, row_user.numel(), n_i, N.ptr(ub),
           N.ptr(ib), sb, N.ptr(rblock_chunk), N.ptr(row_user), N.ptr(table), n_u, 0)
    got, want = table.cpu().numpy(), full.cpu().numpy()
    assert np.array_equal(got[keep], want[keep])
    assert np.all(got[~keep] == -123.0)
    # the fixed-capacity layout (trec_topk_rows_collect): [n_sb][rcap] with per-superblock counts, users in any order
    rcap = 1024
    counts = keep.sum(1).astype(np.int32)
    assert counts.max() <= rcap
    ru = np.full((n_sb, rcap), -77, np.int32)
    for s in range(n_sb):
        ru[s, :counts[s]] = rng.permutation(np.nonzero(keep[s])[0])
    table2 = torch.full((n_sb, n_u), -123.0, dtype=torch.float32, device="cuda")
    dcounts, dru = dev(counts), dev(ru.reshape(-1))
    N.call("trec_score_gemm_blockmax_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), d, n_sb * rcap, n_i, N.ptr(ub),
           N.ptr(ib), sb, N.ptr(dcounts), N.ptr(dru), N.ptr(table2), n_u, rcap // 512)
    assert np.array_equal(table2.cpu().numpy(), got)


def test_cascade_hot_superblocks_of_a_popular_catalogue(ops):
    """A Zipf-popular catalogue (what a fitted model looks like): a few items with large norms and biases are wanted by
    nearly every user, so their superblocks are kept by more users than the compaction's fixed capacity although only a few
    percent of ALL pairs are kept.  Those "hot" superblocks are refined for every user by a dense launch over the list
    (trec_topk_rows_hot / trec_score_gemm_blockmax_hot): the cascade itself runs (no fall-back) and stays exact."""
    rng = np.random.default_rng(21)
    n_u, n_i, d, k = 2100, 300_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32) / np.sqrt(d)
    v = rng.standard_normal((n_i, d)).astype(np.float32) / np.sqrt(d)
    pop = rng.permutation(n_i)[:40]                       # 40 popular items, spread over ~40 of the 586 superblocks
    v[pop] *= 3.0
    ub = (0.05 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.05 * rng.standard_normal(n_i)).astype(np.float32)
    ib[pop] += 2.0
    from tensorrec_amd import _native as N
    ops.FILTER_DEBUG = {}
    N.set_tuning("cascade_rcap_pct", 10)                  # (2,100 users: a list capacity of 10% instead of 50% of them)
    N.set_tuning("cascade_prerefine", 0)                  # (the threshold of round 4: with the pre-refined one fewer users want them)
    try:
        vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
        dbg = dict(ops.FILTER_DEBUG)
        N.set_tuning("cascade_prerefine", 1)              # ... and with the pre-refinement: the same lists
        vals2, idx2, stats2, _, _ = run_cascade(ops, u, v, k, ub, ib)
        dbg2 = dict(ops.FILTER_DEBUG)
    finally:
        ops.FILTER_DEBUG = None
        N.set_tuning("cascade_rcap_pct", int(100 * ops.CASCADE_ROW_CAPACITY))
        N.set_tuning("cascade_prerefine", 1)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert np.array_equal(idx2, ri) and np.array_equal(vals2, rv)
    assert stats2["prefilter"] == "int8" and dbg2["prerefine_users_ok"] >= n_u // 2, (stats2, dbg2)
    assert dbg2["int8_pairs_wanted"] <= dbg["int8_pairs_wanted"], (dbg, dbg2)      # the sharper threshold never keeps more
    assert stats["prefilter"] == "int8", stats            # the cascade ran: hot rows did not overflow it
    assert dbg["hot_superblocks"] >= 3, dbg               # ... and there were hot rows (wanted by > 1,024 of the 2,100 users)
    assert dbg["int8_pairs_wanted"] < 0.2 * dbg["int8_pairs_total"]


def test_hot_launch_skips_the_users_the_pre_refinement_listed(ops):
    """Pre-refinement AND hot rows at once: 12 popular items are everybody's k best superblocks, so the
    pre-refining launch lists them for as many users as a superblock's list holds (512 here) and the compaction finds the same
    superblocks wanted by the ~1,200 others -- hot.  The dense launch over the hot superblocks skips the users whose entry the
    pre-refinement marked (-inf): nobody's candidates are listed twice, so the lists are never longer than without the
    pre-refinement (same provisional floor there, a sharper one elsewhere).  Exact either way."""
    rng = np.random.default_rng(22)
    n_u, n_i, d, k = 2100, 300_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32) / np.sqrt(d)
    v = rng.standard_normal((n_i, d)).astype(np.float32) / np.sqrt(d)
    pop = rng.permutation(n_i)[:12]
    v[pop] *= 3.0
    ub = (0.05 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.05 * rng.standard_normal(n_i)).astype(np.float32)
    ib[pop] += 2.0
    from tensorrec_amd import _native as N
    ops.FILTER_DEBUG = {}
    stats_were = ops.CANDIDATE_STATS
    ops.CANDIDATE_STATS = True
    N.set_tuning("cascade_rcap_pct", 0)                   # (a list capacity of 512 users per superblock)
    try:
        N.set_tuning("cascade_prerefine", 0)
        vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
        N.set_tuning("cascade_prerefine", 1)
        vals2, idx2, stats2, _, _ = run_cascade(ops, u, v, k, ub, ib)
        dbg2 = dict(ops.FILTER_DEBUG)
    finally:
        ops.FILTER_DEBUG = None
        ops.CANDIDATE_STATS = stats_were
        N.set_tuning("cascade_rcap_pct", int(100 * ops.CASCADE_ROW_CAPACITY))
        N.set_tuning("cascade_prerefine", 1)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert np.array_equal(idx2, ri) and np.array_equal(vals2, rv)
    assert stats2["prefilter"] == "int8" and stats2["tail"] == "candidate lists", stats2
    assert dbg2["prerefine_listed"] == 1 and dbg2["prerefine_pairs"] > 0 and dbg2["hot_superblocks"] >= 1, dbg2
    assert stats2["candidates_per_user"] <= stats["candidates_per_user"] + 1e-9, (stats, stats2)


def test_rows_hot_lists_rows_over_capacity(ops):
    """trec_topk_rows_hot: rows with count > rcap, ascending, -1 padded; their counts zeroed; status = {rows, overflow, hot rows}."""
    from tensorrec_amd import _native as N
    counts = np.array([5, 700, 512, 513, 0, 9000, 100, 513], dtype=np.int32)
    rcap, n_users = 512, 1000
    for hot_cap, max_pairs, over in ((8, 1 << 40, 0), (3, 1 << 40, 1), (8, 4616, 1), (8, 4617, 0)):
        rc = dev(counts.copy())
        hot = torch.full((hot_cap,), -7, dtype=torch.int32, device="cuda")
        status = torch.zeros((3,), dtype=torch.int64, device="cuda")
        N.call("trec_topk_rows_hot", N.ptr(rc), len(counts), rcap, n_users, N.ptr(hot), hot_cap, max_pairs, N.ptr(status))
        want = [1, 3, 5, 7]
        assert hot.cpu().tolist() == (want + [-1] * hot_cap)[:hot_cap]
        assert rc.cpu().tolist() == [5, 0, 512, 0, 0, 0, 100, 0]
        rows = 512 + 512 + 0 + 512 + min(len(want), hot_cap) * 1024
        assert status.cpu().tolist() == [rows, over, len(want)]


@pytest.mark.parametrize("cand_cap", [128, 6])
def test_one_pass_scan_equals_select_plus_collect(ops, cand_cap):
    """Behind the int8 stage, select + collect run as ONE pass over big tables (trec_topk_scan_blocks with a provisional floor
    from tau8 + trec_topk_prune_candidates); forced here on a small one.  cand_cap = 6: nearly every user has more entries
    above the provisional floor than its candidate list holds -- those are re-collected by the masked pass
    (trec_topk_collect_blocks_masked).  Same result as the two-pass form and as the oracle."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(41)
    n_u, n_i, d, k = 1300, 300_000, 128, 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.3 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.3 * rng.standard_normal(n_i)).astype(np.float32)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    old_cap = ops.FILTER_CANDIDATES
    try:
        ops.FILTER_CANDIDATES = cand_cap
        N.set_tuning("filter_scan_one_pass", 2)
        N.set_tuning("cascade_candidates", 0)             # (the table-driven tail: the default goes through candidate lists)
        vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
    finally:
        ops.FILTER_CANDIDATES = old_cap
        N.set_tuning("filter_scan_one_pass", 1)
        N.set_tuning("cascade_candidates", 1)
    assert stats["prefilter"] == "int8" and "tail" not in stats
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert stats["flagged_users"] <= 13


@pytest.mark.parametrize("d,n_u,n_i,k,kind", [(128, 1100, 300_000, 10, "gauss"), (64, 700, 280_000, 7, "gauss"),
                                              (128, 900, 270_000, 16, "clustered"), (128, 600, 40_000, 10, "gauss")])
def test_prerefined_threshold_keeps_fewer_pairs_and_the_same_lists(ops, d, n_u, n_i, k, kind):
    """The pre-refinement (csrc/topk_filter.hip): the k superblocks with a user's k largest int8 lower bounds are refined
    first and tau = max(tau8, min of their bf16 maxima - eps).  With it and without it the lists are the oracle's, bit for bit;
    with it the compaction keeps fewer (superblock, user) pairs."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(d + n_u + k)
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    if kind == "clustered":                               # fitted-like rows: cluster directions, heterogeneous norms, popular items
        cu = rng.standard_normal((32, d)).astype(np.float32)
        u = (0.6 * u + cu[rng.integers(0, 32, n_u)]) * rng.uniform(0.3, 2.0, (n_u, 1)).astype(np.float32)
        v = (0.6 * v + cu[rng.integers(0, 32, n_i)]) * rng.uniform(0.5, 1.5, (n_i, 1)).astype(np.float32)
    else:
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    ub = (0.05 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.05 * rng.standard_normal(n_i)).astype(np.float32)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    wanted = {}
    try:
        for flag in (0, 1):
            N.set_tuning("cascade_prerefine", flag)
            ops.FILTER_DEBUG = {}
            vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
            dbg = dict(ops.FILTER_DEBUG)
            assert np.array_equal(idx, ri) and np.array_equal(vals, rv), (flag, stats)
            if str(stats.get("prefilter")) == "int8":
                wanted[flag] = dbg["int8_pairs_wanted"]
                if flag:
                    assert dbg["prerefine_users_ok"] >= n_u // 2, dbg
    finally:
        ops.FILTER_DEBUG = None
        N.set_tuning("cascade_prerefine", 1)
    if len(wanted) == 2:
        assert wanted[1] <= wanted[0], wanted
        if kind == "gauss" and n_i >= 262_144:
            assert wanted[1] < 0.8 * wanted[0], wanted     # (Gaussian rows: about half; the k pre-refined superblocks count as kept)


@pytest.mark.parametrize("d,biased,n_u,n_i", [(128, True, 5000, 300_000), (64, False, 3000, 280_000 + 77), (128, True, 700, 1_000_000)])
def test_cascade_with_the_item_resident_refining_kernel(ops, d, biased, n_u, n_i):
    """tuning refine_resident = 1: the refining launches keep a superblock's 512 items in registers and stream its user list in
    segments (csrc/refine_resident.hip) -- same maxima, same candidate lists, so the same exact result as the oracle's
    tf.matmul + tf.nn.top_k (prediction_graphs.py:49-50, recommendation_graphs.py:80); segments of 64 users make every superblock
    span several workgroups, the last catalogue ends inside a superblock."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(d + n_u)
    k = 10
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.1 * rng.standard_normal(n_u)).astype(np.float32) if biased else None
    ib = (0.1 * rng.standard_normal(n_i)).astype(np.float32) if biased else None
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    try:
        for seg in (2048, 64):
            N.set_tuning("refine_resident", 1)
            N.set_tuning("refine_resident_seg", seg)
            vals, idx, stats, _, _ = run_cascade(ops, u, v, k, ub, ib)
            assert np.array_equal(idx, ri) and np.array_equal(vals, rv), (seg, stats)
            assert stats["prefilter"] == "int8" and stats["tail"] == "candidate lists" and stats["flagged_users"] <= n_u // 20, stats
    finally:
        N.set_tuning("refine_resident", 0)
        N.set_tuning("refine_resident_seg", 2048)
