"""Skewed (Zipf-shaped) data: the chunked gathers of csrc/spmm_split.hip, the structured Euclidean pair gradient, the
sliced column sums and the split-K GEMM -- the kernels a MovieLens-20M-shaped fit (BASELINE.json configs[4]) leans on.

Bars: rows at or below the split threshold are the CSR-order fmaf chain bit for bit (== trec_spmm_csr == the oracle);
longer rows are sums of chunk chains added in chunk order: deterministic (two runs agree bit for bit), and within 1e-5
relative of the float64 result."""
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


@pytest.fixture
def small_split():
    """split threshold 16 (chunks of 8) so that small matrices have long rows"""
    from tensorrec_amd import _native
    _native.set_tuning("split_t", 16)
    yield 16
    _native.set_tuning("split_t", 2048)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def skewed_csr(n_rows, n_cols, seed, long_rows=(0, 7, 41), long_nnz=(17, 100, 1000)):
    """short random rows, a few empty, and some rows far above the (small) split threshold"""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for r in range(n_rows):
        n = int(rng.integers(0, 12))
        if r in long_rows:
            n = min(n_cols, long_nnz[long_rows.index(r)])
        if r in (3, n_rows - 1):
            n = 0
        c = rng.choice(n_cols, size=n, replace=False)
        rows += [r] * n
        cols += list(c)
        vals += list(rng.standard_normal(n).astype(np.float32))
    m = sp.csr_matrix((np.array(vals, np.float32), (rows, cols)), shape=(n_rows, n_cols))
    m.sort_indices()
    return m


@pytest.mark.parametrize("d", [4, 64, 128, 256, 1024, 100, 1, 5, 10, 50, 250])
def test_split_spmm_short_rows_exact_long_rows_close_and_deterministic(ops, small_split, d):
    rng = np.random.default_rng(d)
    x = skewed_csr(120, 1500, seed=d)
    w = rng.standard_normal((1500, d)).astype(np.float32)
    indptr, idx, val = dev(x.indptr.astype(np.int64)), dev(x.indices.astype(np.int32)), dev(x.data)
    wt = dev(w)
    got, rs = ops.spmm_split(indptr, idx, val, None, 120, x.nnz, wt, want_rowsum=True)
    again, rs2 = ops.spmm_split(indptr, idx, val, None, 120, x.nnz, wt, want_rowsum=True)
    assert torch.equal(got, again) and torch.equal(rs, rs2)                      # no scheduling dependence
    got, rs = got.cpu().numpy(), rs.cpu().numpy()
    exact = O.spmm_exact(x, w)
    nnz_row = np.diff(x.indptr)
    short = nnz_row <= small_split
    assert (~short).sum() == 3
    assert np.array_equal(got[short], exact[short])                              # the fmaf chain of trec_spmm_csr
    ref = (x.astype(np.float64) @ w.astype(np.float64))
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-4)
    assert np.allclose(rs, np.asarray(x.sum(axis=1)).reshape(-1), rtol=1e-5, atol=1e-5)
    # accumulate: out += result, rowsum += sums
    base = rng.standard_normal((120, d)).astype(np.float32)
    out, rsum = dev(base), dev(np.ones(120, np.float32))
    ops.spmm_split(indptr, idx, val, None, 120, x.nnz, wt, accumulate=True, out=out, want_rowsum=rsum)
    assert np.allclose(out.cpu().numpy(), base + ref, rtol=1e-5, atol=1e-4)
    assert np.allclose(rsum.cpu().numpy(), 1.0 + np.asarray(x.sum(axis=1)).reshape(-1), rtol=1e-5, atol=1e-5)


def test_split_spmm_permuted_values_packed_entries_and_own(ops, small_split):
    rng = np.random.default_rng(5)
    d = 128
    x = skewed_csr(90, 700, seed=9, long_nnz=(17, 100, 700))
    w = rng.standard_normal((700, d)).astype(np.float32)
    own = rng.standard_normal((90, d)).astype(np.float32)
    indptr, idx = dev(x.indptr.astype(np.int64)), dev(x.indices.astype(np.int32))
    perm = rng.permutation(x.nnz).astype(np.int32)
    scrambled = np.empty(x.nnz, np.float32)
    scrambled[perm] = x.data                                                     # entry j carries values[perm[j]]
    a = ops.spmm_split(indptr, idx, dev(x.data), None, 90, x.nnz, dev(w))
    b = ops.spmm_split(indptr, idx, dev(scrambled), dev(perm), 90, x.nnz, dev(w))
    entries = np.stack([x.indices.astype(np.int32), x.data.view(np.int32)], axis=1)
    c = ops.spmm_split(indptr, None, None, None, 90, x.nnz, dev(w), packed=dev(entries))
    assert torch.equal(a, b) and torch.equal(a, c)
    # own: sum_j v_j (own[row] - w[col_j])
    got = ops.spmm_split(indptr, idx, dev(x.data), None, 90, x.nnz, dev(w), own=dev(own)).cpu().numpy()
    rowsum = np.asarray(x.astype(np.float64).sum(axis=1)).reshape(-1, 1)
    l1 = np.asarray(abs(x).astype(np.float64).sum(axis=1)).reshape(-1, 1)
    ref = rowsum * own.astype(np.float64) - x.astype(np.float64) @ w.astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-5 * (l1 * 8).max()


def test_split_spmv_and_segment_sums(ops, small_split):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(2)
    x = skewed_csr(150, 900, seed=1, long_nnz=(17, 200, 900))
    beta = rng.standard_normal(900).astype(np.float32)
    indptr, idx, val = dev(x.indptr.astype(np.int64)), dev(x.indices.astype(np.int32)), dev(x.data)
    short = np.diff(x.indptr) <= small_split
    for b in (dev(beta), None):
        got = ops.spmv_raw(indptr, idx, val, None, 150, x.nnz, b, long_rows=True)
        assert torch.equal(got, ops.spmv_raw(indptr, idx, val, None, 150, x.nnz, b, long_rows=True))
        plain = ops.spmv_raw(indptr, idx, val, None, 150, x.nnz, b, long_rows=False).cpu().numpy()
        got = got.cpu().numpy()
        assert np.array_equal(got[short], plain[short])                          # the same chains as trec_spmv_csr
        ref = x.astype(np.float64) @ beta.astype(np.float64) if b is not None else \
            np.asarray(x.astype(np.float64).sum(axis=1)).reshape(-1)
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-4)
    assert N.query("trec_csr_split_workspace_bytes", -1, 4) == -1


def test_default_threshold_leaves_ordinary_matrices_bit_exact(ops):
    """split threshold 2048: a matrix without long rows goes through the split entry point unchanged, bit for bit"""
    rng = np.random.default_rng(0)
    x = sp.random(300, 200, density=0.2, random_state=3, dtype=np.float32, format="csr")
    w = rng.standard_normal((200, 64)).astype(np.float32)
    got = ops.spmm_split(dev(x.indptr.astype(np.int64)), dev(x.indices.astype(np.int32)), dev(x.data), None, 300, x.nnz,
                         dev(w)).cpu().numpy()
    assert np.array_equal(got, O.spmm_exact(x, w))


def test_transposed_feature_gradient_with_a_long_column(ops):
    """dW = X^T . dOut where one indicator column holds 5000 rows (> 2048): the autograd path picks the chunked gather"""
    from tensorrec_amd.sparse import SparseFeatures
    rng = np.random.default_rng(4)
    n, f, d = 6000, 50, 64
    rows = np.concatenate([np.arange(n), rng.choice(n, 5000, replace=False)])
    cols = np.concatenate([rng.integers(1, f, n), np.zeros(5000, np.int64)])
    x = sp.csr_matrix((rng.standard_normal(rows.size).astype(np.float32), (rows, cols)), shape=(n, f))
    x.sum_duplicates()
    feats = SparseFeatures(x, "cuda")
    assert feats.max_col_nnz >= 5000 and feats.max_row_nnz <= 2
    w = dev(rng.standard_normal((f, d)).astype(np.float32)).requires_grad_(True)
    beta = dev(rng.standard_normal(f).astype(np.float32)).requires_grad_(True)
    g = rng.standard_normal((n, d)).astype(np.float32)
    gb = rng.standard_normal(n).astype(np.float32)
    ops.sparse_dense_matmul(feats, w).backward(dev(g))
    ops.sparse_matvec(feats, beta).backward(dev(gb))
    ref = x.T.astype(np.float64) @ g.astype(np.float64)
    assert np.allclose(w.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-3)
    assert np.allclose(beta.grad.cpu().numpy(), x.T.astype(np.float64) @ gb.astype(np.float64), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("d", [32, 128, 256, 10, 50])
@pytest.mark.parametrize("layout", ["interactions", "samples"])
def test_euclidean_pair_gradient_structured_equals_autograd(ops, d, layout):
    """The structured Euclidean backward (coefficients + (own - other) gathers) against torch autograd on the CPU and
    against the atomic kernel; one popular item holds more pairs than the split threshold."""
    from tensorrec_amd.sparse import Interactions, PairIndex
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(d)
    nu, ni = 3000, 40
    u = rng.standard_normal((nu, d)).astype(np.float32)
    v = rng.standard_normal((ni, d)).astype(np.float32)
    v[5] = u[7]                                                   # a clamped pair (distance 0): no gradient
    ub, ib = rng.standard_normal(nu).astype(np.float32), rng.standard_normal(ni).astype(np.float32)
    if layout == "interactions":
        m = sp.random(nu, ni, density=0.1, random_state=1, dtype=np.float32, format="lil")
        m[:, 0] = 1.0                                             # item 0: 3000 pairs > 2048
        m[7, 5] = 1.0
        m = sp.csr_matrix(m)
        inter = Interactions(m, nu, ni, "cuda")
        assert inter.max_col_nnz == nu
        xu_np, xi_np = inter.x_user.cpu().numpy(), inter.x_item.cpu().numpy()
        xu = PairIndex.make(inter.x_user, inter.x_user32, 0, inter)
        xi = PairIndex.make(inter.x_item, inter.x_item32, 0, inter)
    else:
        S = 8
        items = rng.integers(0, ni, (nu, S)).astype(np.int32)
        items[:, 0] = 0
        items[7, 1] = 5
        flat = dev(items.reshape(-1))
        xu = xi = PairIndex.make(flat, flat, S)
        xu_np, xi_np = np.repeat(np.arange(nu), S), items.reshape(-1).astype(np.int64)
    g = rng.standard_normal(xi_np.size).astype(np.float32)
    ut, vt = dev(u).requires_grad_(True), dev(v).requires_grad_(True)
    ubt, ibt = dev(ub).requires_grad_(True), dev(ib).requires_grad_(True)
    ops.pair_score(ut, vt, xu, xi, ops.MODE_EUCLIDEAN, ubt, ibt).backward(dev(g))
    uc, vc = torch.from_numpy(u).double().requires_grad_(True), torch.from_numpy(v).double().requires_grad_(True)
    ubc, ibc = torch.from_numpy(ub).double().requires_grad_(True), torch.from_numpy(ib).double().requires_grad_(True)
    dist = ((uc[xu_np] - vc[xi_np]) ** 2).sum(1)
    sc = -torch.sqrt(torch.clamp(dist, min=1e-16)) + ubc[xu_np] + ibc[xi_np]
    sc.backward(torch.from_numpy(g).double())
    for a, b in ((ut, uc), (vt, vc), (ubt, ubc), (ibt, ibc)):
        ref = b.grad.numpy()
        assert np.abs(a.grad.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    # the atomic kernel computes the same sums in an arbitrary order
    du, dv = torch.zeros_like(ut), torch.zeros_like(vt)
    u_d, v_d, xu_d, xi_d, g_d = ut.detach(), vt.detach(), dev(xu_np.astype(np.int32)), dev(xi_np.astype(np.int32)), dev(g)
    N.call("trec_pair_score_bwd", N.ptr(u_d), N.ptr(v_d), N.ptr(xu_d), N.ptr(xi_d), N.ptr(g_d), g.size, 0, d,
           ops.MODE_EUCLIDEAN, N.ptr(du), N.ptr(dv), None, None)
    scale = max(1.0, float(dv.abs().max()))
    assert float((du - ut.grad).abs().max()) <= 1e-4 * scale and float((dv - vt.grad).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("shape", [(100000, 1024), (5000, 100), (300, 7), (255, 64)])
def test_colsum_slices(ops, shape):
    rng = np.random.default_rng(shape[1])
    x = rng.standard_normal(shape).astype(np.float32)
    got = ops.colsum(dev(x))
    assert torch.equal(got, ops.colsum(dev(x)))
    ref = x.astype(np.float64).sum(0)
    assert np.allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.sqrt(shape[0]) * 4)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 260, 129), (1000, 256, 5000), (130, 70, 9000), (64, 1024, 33)])
def test_gemm_f32_tiles_edges_and_split_k(ops, ta, tb, M, N, K):
    rng = np.random.default_rng(M + K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    got = ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb)
    assert torch.equal(got, ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb))       # split-K adds in slice order
    assert np.allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.sqrt(K) * 8)


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 260, 129), (1000, 256, 5000), (130, 70, 9000), (64, 1024, 33), (2000, 256, 1003)])
def test_gemm_split_bf16_is_an_fp32_product_to_a_few_1e5(ops, ta, tb, M, N, K):
    """trec_gemm_f32_split_bf16 (the dense-coefficient GEMMs of the tiled WMRB step: gradients of tensorrec.py:487-489, 1e-4 bar):
    both operands as hi + lo bf16 terms, three bf16 MFMAs per block, fp32 accumulation -- within 3e-5 of the largest entry of the
    float64 product on ragged tiles, every transposition, split K, sparse operands and operands of mixed magnitudes (the
    coefficient matrix is 90 % zeros with entries over several decades)."""
    rng = np.random.default_rng(M + K + ta + 2 * tb)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    a *= np.exp(2.0 * rng.standard_normal(a.shape)).astype(np.float32) * (rng.random(a.shape) < 0.3)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    got = ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb, split_bf16=True)
    assert torch.equal(got, ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb, split_bf16=True))
    err = np.abs(got.cpu().numpy() - ref).max()
    assert err <= 3e-5 * np.abs(ref).max(), (err, np.abs(ref).max())
    exact = ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb).cpu().numpy()
    assert np.abs(exact - ref).max() <= err + 1e-5 * np.abs(ref).max()          # (the fp32 product is at least as close)
    # the column sums of the stored A ride along (trans_a: taken by the threads that stage A; otherwise the colsum kernel)
    c2, cs = ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb, split_bf16=True, a_colsum=True)
    assert torch.equal(c2, got)
    ref_cs = a.astype(np.float64).sum(0)
    assert np.allclose(cs.cpu().numpy(), ref_cs, rtol=1e-5, atol=1e-5 * np.abs(a).max() * np.sqrt(a.shape[0]) * 4)


def test_euclid_coef_from_kept_squared_distances_equals_recomputed(ops):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(8)
    nu, ni, d, P = 50, 70, 96, 4000
    u, v = dev(rng.standard_normal((nu, d)).astype(np.float32)), dev(rng.standard_normal((ni, d)).astype(np.float32))
    xu, xi = dev(rng.integers(0, nu, P).astype(np.int32)), dev(rng.integers(0, ni, P).astype(np.int32))
    g = dev(rng.standard_normal(P).astype(np.float32))
    out, sq = torch.empty(P, device="cuda"), torch.empty(P, device="cuda")
    N.call("trec_pair_score_fwd", N.ptr(u), N.ptr(v), N.ptr(xu), N.ptr(xi), P, 0, d, ops.MODE_EUCLIDEAN, None, None,
           N.ptr(out), N.ptr(sq))
    assert torch.equal(out, -torch.sqrt(torch.clamp(sq, min=1e-16)))
    a, b = torch.empty(P, device="cuda"), torch.empty(P, device="cuda")
    N.call("trec_pair_euclid_coef", N.ptr(u), N.ptr(v), N.ptr(xu), N.ptr(xi), N.ptr(g), None, P, 0, d, N.ptr(a))
    N.call("trec_pair_euclid_coef", None, None, None, None, N.ptr(g), N.ptr(sq), P, 0, d, N.ptr(b))
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    ref = -g.double() / torch.sqrt(((u.double()[xu.long()] - v.double()[xi.long()]) ** 2).sum(1))
    assert torch.allclose(b.double(), ref, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n_items", [64, 100, 1682, 4096, 26744, 32768, 32769])
def test_rank_rows_sorted_path_is_bit_exact(ops, n_items):
    """K4 by sorting (rows up to 32768 items) against the counting definition: no ties, a few tied classes (the
    (class, index) list), ties everywhere (marked rows recounted by the counting kernel), -0.0 == +0.0, infinities."""
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(n_items)
    n_users = 5 if n_items > 8192 else 9
    s = rng.standard_normal((n_users, n_items)).astype(np.float32)
    s[1, rng.choice(n_items, 40, replace=False)] = 0.25                    # one class of 40
    s[2] = np.round(s[2] * 2000) / 2000                                    # many small classes
    s[3] = rng.integers(0, 5, n_items).astype(np.float32)                  # five huge classes -> recount
    s[4, :8] = [0.0, -0.0, np.inf, -np.inf, np.inf, -0.0, 0.0, -np.inf]
    got = ops.rank_rows(dev(s)).cpu().numpy()
    assert np.array_equal(got, O.rank_predictions_exact(s))
    N.set_tuning("rank_sorted", 0)
    try:
        assert np.array_equal(ops.rank_rows(dev(s)).cpu().numpy(), got)    # the counting kernel alone agrees
    finally:
        N.set_tuning("rank_sorted", 1)


@pytest.mark.parametrize("n_items", [300, 5000, 70000])
def test_rank_of_pairs_by_user_equals_wave_per_pair(ops, n_items):
    """pairs grouped by user (one pass over a row for all of its targets, item slices added with integer atomics)
    against the wave-per-pair kernel and the counting definition: ties, users without pairs, more than eight targets,
    item sub-ranges with a column offset."""
    rng = np.random.default_rng(n_items)
    nu = 37
    s = rng.integers(0, 60, (nu, n_items)).astype(np.float32)            # ties everywhere
    per_user = rng.integers(0, 25, nu)
    per_user[[0, 5, nu - 1]] = 0
    xu = np.repeat(np.arange(nu), per_user).astype(np.int32)
    xi = np.concatenate([np.sort(rng.choice(n_items, c, replace=False)) for c in per_user] + [np.zeros(0, np.int64)]).astype(np.int32)
    indptr = np.concatenate([[0], np.cumsum(per_user)]).astype(np.int64)
    st, tgt = dev(s), dev(s[xu, xi])
    full = O.rank_predictions_exact(s)[xu, xi]
    got = ops.rank_of_pairs_by_user(st, 0, 0, n_items, dev(indptr), dev(xi), tgt, add_one=True).cpu().numpy()
    assert np.array_equal(got, full)
    assert np.array_equal(got, ops.rank_of_pairs(st, 0, 0, n_items, dev(xu), dev(xi), tgt, add_one=True).cpu().numpy())
    # an item sub-range of a slab that starts at global column 100
    b, e = 100 + n_items // 5, 100 + n_items // 2
    part = ops.rank_of_pairs_by_user(st, 100, b, e, dev(indptr), dev(xi + 100), tgt, add_one=False).cpu().numpy()
    ref = ops.rank_of_pairs(st, 100, b, e, dev(xu), dev(xi + 100), tgt, add_one=False).cpu().numpy()
    assert np.array_equal(part, ref)
    empty = ops.rank_of_pairs_by_user(st, 0, 7, 7, dev(indptr), dev(xi), tgt, add_one=True).cpu().numpy()
    assert (empty == 1).all()


def test_group_pairs_by_item_with_lds_counters(ops):
    """trec_group_pairs_by_item_lds (few buckets, very many pairs: MovieLens-shaped catalogues): per-run counters in LDS, a column
    scan over the runs, placement through LDS cursors -- no global atomic.  The result is a valid grouping: indptr = the prefix of
    the bucket sizes, every valid pair appears exactly once inside its item's range with its own user, runs stay in order inside
    a bucket; negative keys are dropped.  ops.group_pairs_by_item picks the form by itself."""
    import numpy as np
    import torch
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(8)
    n_users, S, n_items = 60_000, 80, 20_011                       # 4.8M pairs
    zipf = 1.0 / np.arange(1, n_items + 1) ** 0.7
    xi = rng.choice(n_items, size=n_users * S, p=zipf / zipf.sum()).astype(np.int32)
    xi[rng.random(xi.size) < 0.01] = -1                            # dropped pairs
    n_pairs = xi.size
    assert N.query("trec_group_pairs_lds_runs", n_pairs, n_items) > 0
    dxi = torch.from_numpy(xi).cuda()
    for explicit_users in (False, True):
        xu = (np.arange(n_pairs) // S).astype(np.int32)
        dxu = torch.from_numpy(xu).cuda() if explicit_users else None
        indptr, users_t, perm_t = ops.group_pairs_by_item(dxu, dxi, S, n_items)
        indptr, users_t, perm_t = indptr.cpu().numpy(), users_t.cpu().numpy(), perm_t.cpu().numpy()
        counts = np.bincount(xi[xi >= 0], minlength=n_items)
        assert np.array_equal(indptr, np.concatenate([[0], np.cumsum(counts)]))
        n_valid = int(indptr[-1])
        perm = perm_t[:n_valid]
        assert np.array_equal(np.sort(perm), np.flatnonzero(xi >= 0))              # every valid pair exactly once
        assert np.array_equal(xi[perm], np.repeat(np.arange(n_items), counts))      # ... inside its item's range
        assert np.array_equal(users_t[:n_valid], xu[perm])                          # ... with its own user
    # against the atomic form: the same buckets as SETS
    N.set_tuning("group_pairs_lds", 0)
    try:
        ind2, users2, perm2 = ops.group_pairs_by_item(None, dxi, S, n_items)
    finally:
        N.set_tuning("group_pairs_lds", 1)
    assert np.array_equal(ind2.cpu().numpy(), indptr)
    p2 = perm2.cpu().numpy()[:n_valid]
    for b in (0, 1, 17, n_items - 1):
        assert set(p2[indptr[b]:indptr[b + 1]].tolist()) == set(perm[indptr[b]:indptr[b + 1]].tolist())


def test_ranked_packed_fill_in_two_levels(ops):
    """trec_group_pairs_by_item_staged (the 1e8 sampled pairs of the data-parallel fit: pairs staged by destination window, placed
    while the window sits in L2): with the ranks given every pair's slot is fixed -- indptr[item] + rank -- so the entries must be
    IDENTICAL to the one-level fill's, bit for bit, and to a NumPy placement; explicit and implicit users."""
    import numpy as np
    import torch
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(21)
    n_users, S, n_items = 3_000, 50, 5_003
    zipf = 1.0 / np.arange(1, n_items + 1) ** 0.9
    xi = rng.choice(n_items, size=n_users * S, p=zipf / zipf.sum()).astype(np.int32)
    n_pairs = xi.size
    vals = rng.standard_normal(n_pairs).astype(np.float32)
    # the rank of a pair inside its item's bucket and the histogram -- what wmrb_user_fused_kernel's atomics hand over
    order = np.argsort(xi, kind="stable")
    counts = np.bincount(xi, minlength=n_items).astype(np.int32)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ranks = np.empty(n_pairs, np.int32)
    ranks[order] = (np.arange(n_pairs) - indptr[xi[order]]).astype(np.int32)
    want = np.zeros((n_pairs, 2), np.int32)
    slot = indptr[xi] + ranks
    want[slot, 0] = (np.arange(n_pairs) // S).astype(np.int32)
    want[slot, 1] = vals.view(np.int32)
    dxi, dr, dv = torch.from_numpy(xi).cuda(), torch.from_numpy(ranks).cuda(), torch.from_numpy(vals).cuda()
    assert N.query("trec_group_pairs_staged_bytes", n_pairs, 12) > 0 and N.query("trec_group_pairs_staged_bytes", n_pairs, 24) == 0

    def run(staged, explicit_users):
        ws32 = torch.zeros((2 * n_items,), dtype=torch.int32, device="cuda")
        ws32[:n_items] = torch.from_numpy(counts).cuda()
        xu = torch.arange(n_pairs, dtype=torch.int32, device="cuda") // S if explicit_users else None
        N.set_tuning("group_pairs_staged", 1 if staged else 0)
        N.set_tuning("group_pairs_staged_min", 0)
        N.set_tuning("group_pairs_window_log2", 12)
        try:
            ind, entries, none = ops.group_pairs_by_item(xu, dxi, S, n_items, workspace_with_counts=ws32, ranks=dr, values=dv)
        finally:
            N.set_tuning("group_pairs_staged", 1)
            N.set_tuning("group_pairs_staged_min", 1 << 24)
            N.set_tuning("group_pairs_window_log2", 22)
        assert none is None
        return ind.cpu().numpy(), entries.cpu().numpy()

    for explicit_users in (False, True):
        ind_a, ent_a = run(True, explicit_users)
        ind_b, ent_b = run(False, explicit_users)
        assert np.array_equal(ind_a, indptr) and np.array_equal(ind_b, indptr)
        assert np.array_equal(ent_a, want)
        assert np.array_equal(ent_b, want)


@pytest.mark.gpu
@pytest.mark.parametrize("two_level,skewed", [(1, False), (0, False), (1, True)])
def test_binned_grouping_without_ranks(ops, two_level, skewed):
    """trec_group_pairs_by_item_binned (the sampled pairs of the 1M x 1M fit: tiles sorted by 4,096-item bin in LDS, then -- two_level,
    the default -- by fine bin, a fine bin's entries placed in LDS and written as one run; two_level = 0: one workgroup slice per bin
    counts / scans / places -- no rank per pair, no global atomic per pair): indptr is the histogram's prefix and every item's
    bucket holds exactly its (user, value) pairs; negative items are skipped; implicit and explicit users; with drop_zero_values the
    pairs whose value is +0 / -0 are left out as well (WMRB samples that violate no margin).  skewed: a third of the pairs fall on
    100 neighbouring items -- their fine bin holds more records than its LDS stage and is placed from cursors instead."""
    import numpy as np
    import torch
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(33)
    n_users, S, n_items = 90_000, 50, 50_003
    xi = rng.integers(0, n_items, size=n_users * S, dtype=np.int32)
    if skewed:
        hot = rng.random(xi.size) < 0.33
        xi[hot] = rng.integers(20_000, 20_100, size=int(hot.sum()), dtype=np.int32)
    xi[rng.integers(0, xi.size, 1000)] = -1                     # skipped pairs
    N.set_tuning("group_pairs_two_level", two_level)
    n_pairs = xi.size
    vals = rng.standard_normal(n_pairs).astype(np.float32)
    zero = rng.random(n_pairs) < 0.6
    vals[zero] = np.where(rng.random(int(zero.sum())) < 0.5, np.float32(0.0), np.float32(-0.0))
    nbytes = int(N.query("trec_group_pairs_binned_bytes", n_pairs, n_items))
    assert nbytes > 0 and int(N.query("trec_group_pairs_binned_bytes", 1000, n_items)) == 0
    assert int(N.query("trec_group_pairs_binned_bytes", n_pairs, 3_000_000)) == 0          # more than 512 bins
    users = (np.arange(n_pairs) // S).astype(np.int32)
    dxi, dv = torch.from_numpy(xi).cuda(), torch.from_numpy(vals).cuda()
    for explicit, drop in ((False, 0), (True, 0), (False, 1), (True, 1)):
        keep = (xi >= 0) & (vals != 0) if drop else xi >= 0
        counts = np.bincount(xi[keep], minlength=n_items)
        indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        xu = torch.from_numpy(users).cuda() if explicit else None
        ind = torch.empty((n_items + 1,), dtype=torch.int64, device="cuda")
        ent = torch.full((n_pairs, 2), -7, dtype=torch.int32, device="cuda")
        ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
        N.call("trec_group_pairs_by_item_binned", N.ptr(xu), N.ptr(dxi), N.ptr(dv), n_pairs, S, n_items, drop, N.ptr(ws),
               nbytes, N.ptr(ind), N.ptr(ent))
        assert np.array_equal(ind.cpu().numpy(), indptr)
        got = ent.cpu().numpy()
        n_valid = int(indptr[-1])
        assert np.all(got[n_valid:] == -7)                       # nothing written past the valid pairs
        item_of_slot = np.repeat(np.arange(n_items), counts)
        order_g = np.lexsort((got[:n_valid, 1], got[:n_valid, 0], item_of_slot))
        ref_item, ref_user, ref_val = xi[keep], users[keep], vals[keep].view(np.int32)
        order_r = np.lexsort((ref_val, ref_user, ref_item))
        assert np.array_equal(item_of_slot[order_g], ref_item[order_r])
        assert np.array_equal(got[:n_valid, 0][order_g], ref_user[order_r])
        assert np.array_equal(got[:n_valid, 1][order_g], ref_val[order_r])
    N.set_tuning("group_pairs_two_level", 1)


@pytest.mark.gpu
def test_fused_wmrb_step_with_the_binned_sort_equals_the_ranked_sort(ops):
    """From 2^22 sampled pairs on, the fused WMRB step groups them by item through the rank-free binned partition (no histogram
    atomics in the fused kernel) instead of histogram ranks + fill: losses, predictions and user-side gradients identical (the user
    side does not depend on the sort), item-side gradients equal up to the summation order inside an item's bucket."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    from tensorrec_amd import _native as N
    from tensorrec_amd.sparse import Interactions
    rng = np.random.default_rng(12)
    n_users, n_items, d, S, per = 45_000, 20_000, 32, 100, 5
    cols = rng.integers(0, n_items, size=(n_users, per), dtype=np.int32)
    m = sp.csr_matrix((np.ones(n_users * per, np.float32), cols.reshape(-1), np.arange(0, (n_users + 1) * per, per, dtype=np.int64)),
                      shape=(n_users, n_items))
    m.sum_duplicates()
    m.data[:] = 1.0
    inter = Interactions(m, n_users, n_items, "cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    u = 0.3 * torch.randn((n_users, d), device="cuda", generator=g)
    v = 0.3 * torch.randn((n_items, d), device="cuda", generator=g)
    ub = 0.1 * torch.randn((n_users,), device="cuda", generator=g)
    ib = 0.1 * torch.randn((n_items,), device="cuda", generator=g)
    samples = ops.sample_items(n_users, n_items, S, False, 0, 1)
    assert n_users * S >= ops.GROUP_BINNED_MIN_PAIRS
    outs, kept = [], []
    try:
        for binned, drop in ((1, 1), (1, 0), (0, 1)):
            N.set_tuning("group_pairs_binned", binned)
            N.set_tuning("group_pairs_drop_zero", drop)
            ops.KERNEL_EVENTS = []
            outs.append([t.cpu().numpy() for t in ops.wmrb_fused_step(u, v, ub, ib, inter, samples)])
            names = {n for n, _, _ in ops.KERNEL_EVENTS}
            ops.KERNEL_EVENTS = None
            assert ("group_pairs_binned" in names) == bool(binned), names
            kept.append(int(ops.LAST_FUSED_STATS["sampled_pairs_kept"].item()))
    finally:
        ops.KERNEL_EVENTS = None
        N.set_tuning("group_pairs_binned", 1)
        N.set_tuning("group_pairs_drop_zero", 1)
    # (samples that violate no margin of their user have coefficient 0: left out of the sort by default, nothing else changes)
    assert kept[1] == kept[2] == n_users * S and 0 < kept[0] < kept[1], kept
    lb, pb, dub, dvb, dubb, dibb = outs[2]
    gmax = np.abs(dvb).max()
    for la, pa, dua, dva, duba, diba in outs[:2]:
        assert np.array_equal(la, lb) and np.array_equal(pa, pb) and np.array_equal(dua, dub) and np.array_equal(duba, dubb)
        assert gmax > 0 and np.abs(dva - dvb).max() <= 2e-5 * gmax
        assert np.abs(diba - dibb).max() <= 2e-5 * max(1e-30, np.abs(dibb).max())
