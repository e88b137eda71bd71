"""Kernel-level parity: every HIP entry point (called through the C ABI via tensorrec_amd.ops) against the oracle
on the same seeded inputs.  Bars: integer / index outputs bit-exact; fp32 paths written to the oracle's fmaf order
bit-exact; everything else within the tolerance written next to the assertion (north_star: 1e-4 relative)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import oracle as O
from oracle import device_sampler as DS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tensorrec_amd import ops as _ops, _native
    _native.require_gpu()
    _native.load()
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def feats(m):
    from tensorrec_amd.sparse import SparseFeatures
    return SparseFeatures(m, "cuda")


def rand_csr(n, f, density, seed, empty_rows=True):
    m = sp.random(n, f, density=density, random_state=seed, dtype=np.float32, format="lil")
    if empty_rows and n > 3:
        m[1, :] = 0
        m[n - 1, :] = 0
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    return m


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("d", [4, 64, 100, 128, 256, 1024, 5, 1])
def test_spmm_bit_exact(ops, d):
    rng = np.random.default_rng(d)
    x = rand_csr(203, 57, 0.15, seed=d)
    w = rng.standard_normal((57, d)).astype(np.float32)
    f = feats(x)
    got = ops.spmm_raw(f.indptr, f.indices, f.values, None, f.shape[0], f.nnz, dev(w)).cpu().numpy()
    assert np.array_equal(got, O.spmm_exact(x, w))


@pytest.mark.parametrize("n,d", [(5003, 128), (4096, 64), (7, 4), (1001, 256), (130, 1024)])
def test_spmm_one_per_row_bit_exact(ops, n, d):
    """identity / indicator features (exactly one non-zero per row): the kernel that skips the row pointer."""
    rng = np.random.default_rng(n + d)
    cols = rng.integers(0, 977, n)
    x = sp.csr_matrix((rng.standard_normal(n).astype(np.float32), (np.arange(n), cols)), shape=(n, 977))
    w = rng.standard_normal((977, d)).astype(np.float32)
    f = feats(x)
    assert f.one_per_row
    got = ops.spmm_raw(f.indptr, f.indices, f.values, None, n, f.nnz, dev(w), one_per_row=True).cpu().numpy()
    assert np.array_equal(got, O.spmm_exact(x, w))
    generic = ops.spmm_raw(f.indptr, f.indices, f.values, None, n, f.nnz, dev(w)).cpu().numpy()
    assert np.array_equal(got, generic)
    two = sp.csr_matrix(([1.0, 2.0, 3.0], ([0, 0, 1], [1, 2, 0])), shape=(2, 3), dtype=np.float32)
    assert not feats(two).one_per_row


def test_spmm_identity_rows_batched_path(ops):
    """identity-style features (1 nnz per row) take the 4-rows-per-subgroup path (n_rows >= 4096)."""
    rng = np.random.default_rng(0)
    n, d = 5003, 128
    perm = rng.permutation(n)
    x = sp.csr_matrix((rng.standard_normal(n).astype(np.float32), (np.arange(n), perm)), shape=(n, n))
    w = rng.standard_normal((n, d)).astype(np.float32)
    f = feats(x)
    got = ops.spmm_raw(f.indptr, f.indices, f.values, None, n, f.nnz, dev(w)).cpu().numpy()
    assert np.array_equal(got, O.spmm_exact(x, w))
    # short ragged rows (0..4 nnz) on the same path
    x2 = rand_csr(6000, 300, 0.004, seed=7)
    w2 = rng.standard_normal((300, 64)).astype(np.float32)
    f2 = feats(x2)
    got2 = ops.spmm_raw(f2.indptr, f2.indices, f2.values, None, 6000, f2.nnz, dev(w2)).cpu().numpy()
    assert np.array_equal(got2, O.spmm_exact(x2, w2))


@pytest.mark.parametrize("d", [64, 100, 6])
def test_spmm_transposed_is_weight_gradient(ops, d):
    """dW = X^T . G through the transposed CSR (val_perm) == oracle SpMM on the explicitly transposed matrix."""
    rng = np.random.default_rng(11)
    x = rand_csr(150, 40, 0.2, seed=3)
    g = rng.standard_normal((150, d)).astype(np.float32)
    f = feats(x)
    indptr_t, rows_t, perm_t = f.transposed()
    got = ops.spmm_raw(indptr_t, rows_t, f.values, perm_t, 40, f.nnz, dev(g)).cpu().numpy()
    assert np.array_equal(got, O.spmm_exact(sp.csr_matrix(x.T), g))


def test_spmm_epilogues(ops):
    rng = np.random.default_rng(5)
    x = rand_csr(97, 33, 0.3, seed=5)
    for d in (64, 100, 6):
        w = rng.standard_normal((33, d)).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        f = feats(x)
        y, inv = ops.spmm_raw(f.indptr, f.indices, f.values, None, 97, f.nnz, dev(w), epilogue=ops.EPI_L2NORM,
                              want_inv=True)
        ref = O.l2_normalize_rows(O.spmm_exact(x, w))
        assert np.allclose(y.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)          # reduction order differs
        r = ops.spmm_raw(f.indptr, f.indices, f.values, None, 97, f.nnz, dev(w), col_bias=dev(b),
                         epilogue=ops.EPI_BIAS_RELU).cpu().numpy()
        assert np.array_equal(r, np.maximum(O.spmm_exact(x, w) + b, np.float32(0)))
    # empty rows normalise to 0 (sum = 0 < eps), not NaN
    assert np.isfinite(y.cpu().numpy()).all()


def test_spmm_accumulate(ops):
    rng = np.random.default_rng(6)
    x = rand_csr(50, 20, 0.3, seed=6)
    w = rng.standard_normal((20, 32)).astype(np.float32)
    base = rng.standard_normal((50, 32)).astype(np.float32)
    f = feats(x)
    out = dev(base.copy())
    ops.spmm_raw(f.indptr, f.indices, f.values, None, 50, f.nnz, dev(w), accumulate=True, out=out)
    assert np.allclose(out.cpu().numpy(), base + O.spmm_exact(x, w), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("d", [128, 6])
def test_spmm_rowsum_epilogue(ops, d):
    """epilogue 3: the gather result as epilogue 0 (bit-identical) plus the row sums of the (permuted) values, with and
    without accumulation -- what the item-bias gradient rides on (vec4 path and scalar fallback)."""
    from tensorrec_amd import _native as N
    m = rand_csr(300, 90, 0.2, seed=d)
    f = feats(m)
    rng = np.random.default_rng(d)
    w = dev(rng.standard_normal((90, d)).astype(np.float32))
    perm = rng.permutation(m.nnz).astype(np.int32)
    vals = rng.standard_normal(m.nnz).astype(np.float32)
    dvals, dperm = dev(vals), dev(perm)                 # kept alive: N.ptr() takes raw pointers
    base = ops.spmm_raw(f.indptr, f.indices, dvals, dperm, 300, m.nnz, w)
    out = torch.empty_like(base)
    rs = torch.empty((300,), dtype=torch.float32, device="cuda")
    N.call("trec_spmm_csr", N.ptr(f.indptr), N.ptr(f.indices), N.ptr(dvals), N.ptr(dperm), 300, m.nnz, N.ptr(w), d,
           None, 3, 0, N.ptr(out), N.ptr(rs))
    assert torch.equal(out, base)
    ref = np.array([vals[perm[m.indptr[r]:m.indptr[r + 1]]].astype(np.float64).sum() for r in range(300)])
    assert np.allclose(rs.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    N.call("trec_spmm_csr", N.ptr(f.indptr), N.ptr(f.indices), N.ptr(dvals), N.ptr(dperm), 300, m.nnz, N.ptr(w), d,
           None, 3, 1, N.ptr(out), N.ptr(rs))
    assert np.allclose(out.cpu().numpy(), 2 * base.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(rs.cpu().numpy(), 2 * ref, rtol=1e-5, atol=1e-5)


def test_project_biases_golden(ops, goldens):
    g = goldens["project_biases"]
    f = feats(g["features"])
    got = ops.sparse_matvec(f, dev(g["feature_biases"].astype(np.float32))).cpu().numpy()
    assert (got == g["expected_result"]).all()


def test_sparse_to_dense(ops):
    x = rand_csr(31, 17, 0.3, seed=9)
    assert np.array_equal(ops.sparse_to_dense(feats(x)).cpu().numpy(), x.toarray())


def test_row_l2norm_fwd_bwd(ops):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((77, 100)).astype(np.float32)
    x[3] = 0                                                       # clamped row
    g = rng.standard_normal((77, 100)).astype(np.float32)
    xt = dev(x).requires_grad_(True)
    y = ops.l2_normalize_rows(xt)
    y.backward(dev(g))
    xc = torch.from_numpy(x).requires_grad_(True)
    yc = xc * torch.rsqrt(torch.clamp((xc * xc).sum(1, keepdim=True), min=1e-12))
    yc.backward(torch.from_numpy(g))
    assert np.allclose(y.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert np.allclose(xt.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False)])
def test_gemm_f32(ops, ta, tb):
    rng = np.random.default_rng(4)
    M, N, K = 150, 70, 90
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    got = ops.gemm_raw(dev(a), dev(b), trans_a=ta, trans_b=tb).cpu().numpy()
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------------ K2
def _uv(nu, ni, d, seed, integer=False):
    rng = np.random.default_rng(seed)
    if integer:      # small integers: every dot product is exact and ties are everywhere
        return (rng.integers(-2, 3, (nu, d)).astype(np.float32), rng.integers(-2, 3, (ni, d)).astype(np.float32))
    return rng.standard_normal((nu, d)).astype(np.float32), rng.standard_normal((ni, d)).astype(np.float32)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("d", [128, 100, 64, 32, 256, 5])
def test_score_store_fp32_bit_exact(ops, d, variant):
    """fp32 MFMA = k-ordered fmaf chain: the whole score matrix (with both biases) equals the C oracle bit for bit."""
    u, v = _uv(301, 777, d, seed=d)
    rng = np.random.default_rng(1)
    ub, ib = rng.standard_normal(301).astype(np.float32), rng.standard_normal(777).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    got = ops.score_store(u_op, v_op, ops.DTYPE_F32, kpad, dev(ub), dev(ib), variant=variant).cpu().numpy()
    ref = O.score_dense_exact(u, v, ub, ib)
    assert np.array_equal(got, ref), "max abs diff %g" % np.abs(got - ref).max()
    got2 = ops.score_store(u_op, v_op, ops.DTYPE_F32, kpad, variant=variant).cpu().numpy()
    assert np.array_equal(got2, O.score_dense_exact(u, v))


def test_score_store_is_not_transposed(ops):
    """asymmetric operands: row/col swap or fragment mix-ups cannot cancel out."""
    u = np.zeros((40, 32), np.float32)
    v = np.zeros((70, 32), np.float32)
    u[np.arange(40), np.arange(40) % 32] = np.arange(1, 41)
    v[np.arange(70), (np.arange(70) * 7) % 32] = np.arange(1, 71) * 0.5
    for dt in (ops.DTYPE_F32, ops.DTYPE_BF16):
        u_op, _, kpad = ops.score_prep(dev(u), dt)
        v_op, _, _ = ops.score_prep(dev(v), dt)
        got = ops.score_store(u_op, v_op, dt, kpad).cpu().numpy()
        assert np.array_equal(got, u @ v.T)          # all values exactly representable in bf16


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("d", [128, 100, 64, 32, 256])
def test_score_store_bf16(ops, d, variant):
    """bf16 operands, fp32 accumulate: compare with the fp32 product of the bf16-ROUNDED inputs (isolates the
    kernel from the rounding of the inputs).  Tolerance 1e-5 * |u||v| (accumulation order only)."""
    u, v = _uv(257, 513, d, seed=d + 1)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    ur = torch.from_numpy(u).bfloat16().float().numpy()
    vr = torch.from_numpy(v).bfloat16().float().numpy()
    assert np.array_equal(u_op.float().cpu().numpy()[:, :d], ur)      # round-to-nearest-even, same as torch
    got = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, variant=variant).cpu().numpy()
    ref = ur.astype(np.float64) @ vr.astype(np.float64).T
    scale = np.linalg.norm(ur, axis=1)[:, None] * np.linalg.norm(vr, axis=1)[None, :]
    assert (np.abs(got - ref) <= 1e-5 * scale + 1e-6).all()
    # and against the true fp32 scores: bf16 inputs cost ~2^-8 relative per element (documented in DESIGN.md)
    ref32 = u.astype(np.float64) @ v.astype(np.float64).T
    assert (np.abs(got - ref32) <= 1e-2 * scale).all()


def test_score_store_cosine_and_euclid(ops):
    u, v = _uv(130, 210, 48, seed=3)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, normalize=True)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32, normalize=True)
    got = ops.score_store(u_op, v_op, ops.DTYPE_F32, kpad).cpu().numpy()
    assert np.allclose(got, O.cosine_dense(u, v), rtol=1e-5, atol=1e-6)
    u_op, u_sq, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, want_sqnorm=True)
    v_op, v_sq, _ = ops.score_prep(dev(v), ops.DTYPE_F32, want_sqnorm=True)
    got = ops.score_store(u_op, v_op, ops.DTYPE_F32, kpad, mode=ops.MODE_EUCLIDEAN, user_sq=u_sq, item_sq=v_sq)
    ref = O.score_dense_euclid_exact(u, v, u_sq.cpu().numpy(), v_sq.cpu().numpy())
    assert np.array_equal(got.cpu().numpy(), ref)          # same squared norms in -> bit-exact
    assert np.allclose(ref, O.euclid_dense(u, v), rtol=1e-4, atol=1e-4)


def test_goldens_on_gpu_prediction_graphs(goldens):
    """The reference's own known-answer tests (test/test_prediction_graphs.py:35-192) through the HIP path."""
    from tensorrec_amd.prediction_graphs import (DotProductPredictionGraph, CosineSimilarityPredictionGraph,
                                                 EuclideanSimilarityPredictionGraph)
    for kind, cls in (("dot", DotProductPredictionGraph), ("cosine", CosineSimilarityPredictionGraph),
                      ("euclidean", EuclideanSimilarityPredictionGraph)):
        g = goldens[kind + "_dense"]
        a1, a2 = dev(g["array_1"].astype(np.float32)), dev(g["array_2"].astype(np.float32))
        got = cls().connect_dense_prediction_graph(a1, a2).cpu().numpy()
        assert np.allclose(got, g["expected_result"], atol=1e-6), kind       # float32 resolution, see oracle test
        g = goldens[kind + "_serial"]
        xu, xi = dev(g["x_user"].astype(np.int64)), dev(g["x_item"].astype(np.int64))
        got = cls().connect_serial_prediction_graph(a1, a2, xu, xi).cpu().numpy()
        assert np.allclose(got, g["expected_result"], atol=1e-6), kind


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("k,chunks", [(10, 1), (16, 4), (1, 2)])
def test_score_topk_fp32_exact(ops, k, chunks, variant):
    """Fused top-k == first k entries of the reference's first tf.nn.top_k (value desc, index asc), bit-exact,
    including ties (integer operands make thousands of exact ties)."""
    for integer in (False, True):
        u, v = _uv(300, 1999, 64, seed=k, integer=integer)
        rng = np.random.default_rng(9)
        ub, ib = rng.standard_normal(300).astype(np.float32), rng.standard_normal(1999).astype(np.float32)
        if integer:
            ub, ib = np.round(ub), np.round(ib)
        u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
        v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
        vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_F32, kpad, k, dev(ub), dev(ib), n_chunks=chunks,
                                   variant=variant)
        rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
        assert np.array_equal(idx.cpu().numpy(), ri), "integer=%s" % integer
        assert np.array_equal(vals.cpu().numpy(), rv)


def test_score_topk_fewer_items_than_k_and_index_base(ops):
    u, v = _uv(70, 7, 32, seed=2)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_F32, kpad, 10, item_index_base=1000)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v), 10)
    ri = np.where(ri >= 0, ri + 1000, -1)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)


@pytest.mark.parametrize("variant", [0, 1])
def test_score_topk_bf16_matches_bf16_scores(ops, variant):
    """bf16 mode: the fused top-k must be the exact top-k OF THE bf16 SCORES the store epilogue produces."""
    u, v = _uv(260, 3001, 128, seed=5)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    scores = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, variant=variant).cpu().numpy()
    vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_BF16, kpad, 10, n_chunks=2, variant=variant)
    rv, ri = O.topk_rows(scores, 10)
    assert np.array_equal(vals.cpu().numpy(), rv)
    assert np.array_equal(idx.cpu().numpy(), ri)


def test_score_topk_euclid_and_cosine(ops):
    u, v = _uv(100, 900, 40, seed=8)
    u_op, u_sq, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, want_sqnorm=True)
    v_op, v_sq, _ = ops.score_prep(dev(v), ops.DTYPE_F32, want_sqnorm=True)
    vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_F32, kpad, 5, mode=ops.MODE_EUCLIDEAN, user_sq=u_sq, item_sq=v_sq)
    ref = O.score_dense_euclid_exact(u, v, u_sq.cpu().numpy(), v_sq.cpu().numpy())
    rv, ri = O.topk_rows(ref, 5)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)


def test_topk_merge(ops):
    rng = np.random.default_rng(0)
    for n_cand in (7, 64, 200, 1024):
        n_u, k = 37, 10
        vals = rng.integers(0, 6, (n_u, n_cand)).astype(np.float32)          # ties
        idx = np.stack([rng.permutation(5000)[:n_cand] for _ in range(n_u)]).astype(np.int32)
        empty = rng.random((n_u, n_cand)) < 0.2
        idx[empty] = -1
        vals[empty] = -np.inf
        ov, oi = ops.topk_merge(dev(vals), dev(idx), k)
        for r in range(n_u):
            real = [(-(vals[r, j]), idx[r, j]) for j in range(n_cand) if idx[r, j] >= 0]
            real.sort()
            exp_i = [i for _, i in real[:k]] + [-1] * max(0, k - len(real))
            exp_v = [-v for v, _ in real[:k]] + [-np.inf] * max(0, k - len(real))
            assert oi[r].cpu().tolist() == exp_i
            assert ov[r].cpu().tolist() == exp_v


# ------------------------------------------------------------------------------------------------ K3
@pytest.mark.parametrize("d", [64, 100, 5, 256])
@pytest.mark.parametrize("mode", ["dot", "euclidean"])
def test_pair_score_fwd_bwd(ops, d, mode):
    rng = np.random.default_rng(d)
    nu, ni, P = 40, 60, 500
    u, v = _uv(nu, ni, d, seed=d)
    ub, ib = rng.standard_normal(nu).astype(np.float32), rng.standard_normal(ni).astype(np.float32)
    xu, xi = rng.integers(0, nu, P), rng.integers(0, ni, P)
    xu[:3], xi[:3] = 0, 0
    g = rng.standard_normal(P).astype(np.float32)
    m = ops.MODE_DOT if mode == "dot" else ops.MODE_EUCLIDEAN
    ut, vt = dev(u).requires_grad_(True), dev(v).requires_grad_(True)
    ubt, ibt = dev(ub).requires_grad_(True), dev(ib).requires_grad_(True)
    s = ops.pair_score(ut, vt, dev(xu.astype(np.int64)), dev(xi.astype(np.int64)), m, ubt, ibt)
    s.backward(dev(g))
    uc, vc = torch.from_numpy(u).requires_grad_(True), torch.from_numpy(v).requires_grad_(True)
    ubc, ibc = torch.from_numpy(ub).requires_grad_(True), torch.from_numpy(ib).requires_grad_(True)
    if mode == "dot":
        sc = (uc[xu] * vc[xi]).sum(1)
    else:
        sc = -torch.sqrt(torch.clamp(((uc[xu] - vc[xi]) ** 2).sum(1), min=1e-16))
    sc = sc + ubc[xu] + ibc[xi]
    sc.backward(torch.from_numpy(g))
    tol = dict(rtol=1e-5, atol=1e-5)
    assert np.allclose(s.detach().cpu().numpy(), sc.detach().numpy(), **tol)
    if mode == "dot":
        assert np.allclose(s.detach().cpu().numpy(), O.pair_dot_exact(u, v, xu, xi, ub, ib), **tol)
    for a, b in ((ut, uc), (vt, vc), (ubt, ubc), (ibt, ibc)):
        assert np.allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_pair_score_implicit_users(ops):
    """[U, S] sample layout: user of pair p is p // S (util.py:16-19)."""
    from tensorrec_amd.sparse import PairIndex
    rng = np.random.default_rng(1)
    nu, ni, S, d = 13, 50, 7, 32
    u, v = _uv(nu, ni, d, seed=4)
    items = rng.integers(0, ni, (nu, S)).astype(np.int32)
    flat = dev(items.reshape(-1))
    x = PairIndex.make(flat, flat, S)
    got = ops.pair_score(dev(u), dev(v), x, x).cpu().numpy()
    ref = O.dot_serial(u, v, np.repeat(np.arange(nu), S), items.reshape(-1))
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ K4
def test_rank_rows_golden(ops, goldens):
    g = goldens["rank_predictions"]
    got = ops.rank_rows(dev(g["predictions"].astype(np.float32))).cpu().numpy()
    assert got.dtype == np.int32 and (got == g["expected_ranks"]).all()


@pytest.mark.parametrize("n_items", [1, 5, 1024, 1682, 2048, 2049, 7001])
def test_rank_rows_exact(ops, n_items):
    rng = np.random.default_rng(n_items)
    s = rng.standard_normal((9, n_items)).astype(np.float32)
    t = rng.integers(0, 5, (9, n_items)).astype(np.float32)          # heavy ties
    for x in (s, t):
        assert np.array_equal(ops.rank_rows(dev(x)).cpu().numpy(), O.rank_predictions_exact(x))


def test_rank_of_pairs_shards_add_up(ops):
    rng = np.random.default_rng(3)
    nu, ni, P = 11, 3000, 200
    s = rng.integers(0, 50, (nu, ni)).astype(np.float32)
    xu, xi = rng.integers(0, nu, P).astype(np.int32), rng.integers(0, ni, P).astype(np.int32)
    full = O.rank_predictions_exact(s)[xu, xi]
    st = dev(s)
    tgt = dev(s[xu, xi])
    total = torch.zeros(P, dtype=torch.int32, device="cuda")
    bounds = [0, 700, 701, 2048, 3000]
    for b, e in zip(bounds[:-1], bounds[1:]):
        total += ops.rank_of_pairs(st, 0, b, e, dev(xu), dev(xi), tgt, add_one=(b == 0))
    assert np.array_equal(total.cpu().numpy(), full)
    # a slab that only holds columns [700, 3000) with col_offset
    slab = dev(np.ascontiguousarray(s[:, 700:]))
    part = ops.rank_of_pairs(slab, 700, 700, 3000, dev(xu), dev(xi), tgt, add_one=False)
    part0 = ops.rank_of_pairs(st, 0, 0, 700, dev(xu), dev(xi), tgt, add_one=True)
    assert np.array_equal((part + part0).cpu().numpy(), full)


# ------------------------------------------------------------------------------------------------ K6
def _interactions(nu, ni, density, seed):
    from tensorrec_amd.sparse import Interactions
    m = sp.random(nu, ni, density=density, random_state=seed, dtype=np.float32, format="csr")
    m.data = (m.data - 0.3).astype(np.float32)          # mix of positive and negative interactions
    m[2, :] = 0
    m = sp.csr_matrix(m)
    m.eliminate_zeros()
    return m, Interactions(m, nu, ni, "cuda")


@pytest.mark.parametrize("balanced", [False, True])
@pytest.mark.parametrize("S", [5, 64, 300])
def test_wmrb_fwd_bwd(ops, balanced, S):
    nu, ni = 37, 90
    m, inter = _interactions(nu, ni, 0.2, seed=S)
    rng = np.random.default_rng(S)
    pred = rng.standard_normal(m.nnz).astype(np.float32)
    samp = rng.standard_normal((nu, S)).astype(np.float32)
    rows, cols, vals, _ = O.to_coo_like_reference(m)
    if balanced:
        ref = O.balanced_wmrb_loss(pred, rows, cols, vals, samp, ni, S, (nu, ni))
    else:
        ref = O.wmrb_loss(pred, rows, vals, samp, ni, S)
    pt, st = dev(pred).requires_grad_(True), dev(samp).requires_grad_(True)
    loss = ops.wmrb_loss(pt, st, inter, balanced=balanced)
    assert loss.shape == (int((vals > 0).sum()),)
    assert np.allclose(loss.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    go = rng.standard_normal(loss.shape[0]).astype(np.float32)
    loss.backward(dev(go))
    # torch-CPU autograd of the reference formula
    pc, sc = torch.from_numpy(pred).requires_grad_(True), torch.from_numpy(samp).requires_grad_(True)
    mask = torch.from_numpy(vals > 0)
    xu = torch.from_numpy(rows)
    summ = torch.clamp(1.0 - pc[mask][:, None] + sc[xu[mask]], min=0.0)
    smr = (float(ni) / float(S)) * summ.sum(1)
    if balanced:
        pv = torch.from_numpy(vals)[mask]
        per_item = torch.zeros(ni).index_add_(0, torch.from_numpy(cols)[mask], pv)
        smr = smr * pv / per_item[torch.from_numpy(cols)[mask]]
    torch.log(smr + 1.0).backward(torch.from_numpy(go))
    assert np.allclose(pt.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert np.allclose(st.grad.cpu().numpy(), sc.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("S", [1, 63, 100, 256])
def test_wmrb_wave_per_user_kernels_match_workgroup_kernels(ops, S):
    """S <= 256 runs a wave per user; it must reproduce the workgroup-per-user kernels bit for bit (same lane/order
    split), including users without interactions, non-positive interactions and balanced weights."""
    import tensorrec_amd as T
    nu, ni = 133, 70
    m, inter = _interactions(nu, ni, 0.3, seed=S)
    rng = np.random.default_rng(S)
    pred = rng.standard_normal(m.nnz).astype(np.float32)
    samp = rng.standard_normal((nu, S)).astype(np.float32)
    go = None
    outs = []
    for wave in (1, 0):
        T._native.set_tuning("wmrb_wave", wave)
        try:
            for balanced in (False, True):
                pt, st = dev(pred).requires_grad_(True), dev(samp).requires_grad_(True)
                loss = ops.wmrb_loss(pt, st, inter, balanced=balanced)
                if go is None:
                    go = rng.standard_normal(loss.shape[0]).astype(np.float32)
                loss.backward(dev(go))
                outs.append((loss.detach().cpu().numpy(), pt.grad.cpu().numpy(), st.grad.cpu().numpy()))
        finally:
            T._native.set_tuning("wmrb_wave", 1)
    for a, b in zip(outs[:2], outs[2:]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_wmrb_many_positives_per_user(ops):
    """more positives than one LDS pass holds (1024) -> the multi-pass accumulation path."""
    from tensorrec_amd.sparse import Interactions
    nu, ni, S = 3, 2600, 40
    m = sp.csr_matrix(np.ones((nu, ni), np.float32))
    inter = Interactions(m, nu, ni, "cuda")
    rng = np.random.default_rng(0)
    pred = rng.standard_normal(m.nnz).astype(np.float32)
    samp = rng.standard_normal((nu, S)).astype(np.float32)
    pt, st = dev(pred).requires_grad_(True), dev(samp).requires_grad_(True)
    loss = ops.wmrb_loss(pt, st, inter)
    loss.sum().backward()
    rows, cols, vals, _ = O.to_coo_like_reference(m)
    assert np.allclose(loss.detach().cpu().numpy(), O.wmrb_loss(pred, rows, vals, samp, ni, S), rtol=1e-5, atol=1e-6)
    pc, sc = torch.from_numpy(pred).requires_grad_(True), torch.from_numpy(samp).requires_grad_(True)
    summ = torch.clamp(1.0 - pc[:, None] + sc[torch.from_numpy(rows)], min=0.0)
    torch.log((float(ni) / S) * summ.sum(1) + 1.0).sum().backward()
    assert np.allclose(st.grad.cpu().numpy(), sc.grad.numpy(), rtol=1e-4, atol=1e-4)
    assert np.allclose(pt.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n", [1, 7, 5000, 300001])
def test_rmse_fwd_bwd(ops, n):
    rng = np.random.default_rng(n)
    y, p = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    pt = dev(p).requires_grad_(True)
    loss = ops.rmse_loss(pt, dev(y))
    loss.backward()
    ref = O.rmse_loss(p, y)
    assert np.allclose(float(loss), ref, rtol=1e-5)
    pc = torch.from_numpy(p).requires_grad_(True)
    torch.sqrt(torch.mean((torch.from_numpy(y) - pc) ** 2)).backward()
    assert np.allclose(pt.grad.cpu().numpy(), pc.grad.numpy(), rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------------ K7
@pytest.mark.parametrize("replace", [False, True])
def test_sampler_bit_exact_vs_restatement(ops, replace):
    for (nu, ni, S, seed, step) in ((7, 1000, 50, 0x1234567890ABCDEF, 3), (300, 97, 30, 5, 1), (2, 5, 5, 9, 2),
                                    (1000, 1682, 168, 0, 7), (5000, 100_000, 9, 77, 4), (3, 70_000, 2500, 1, 1)):
        # (without replacement and >= 8 samples per user: the keyed kernel -- a workgroup owns 2,048 consecutive samples, up to 228
        # users' keys in LDS; below, and with replacement: one thread per sample)
        got = ops.sample_items(nu, ni, S, replace, seed, step).cpu().numpy()
        assert np.array_equal(got, DS.sample_items(nu, ni, S, replace, seed, step))
        # a user shard draws exactly the rows of the whole-population table
        if nu >= 4:
            part = ops.sample_items(nu - 3, ni, S, replace, seed, step, user_base=3).cpu().numpy()
            assert np.array_equal(part, got[3:])


def test_sampler_contract(ops):
    """util.sample_items contract (util.py:12-21): [U, S], all items reachable, distinct per user without replacement."""
    x = ops.sample_items(4000, 97, 30, False, 123, 1).cpu().numpy()
    assert x.shape == (4000, 30) and x.min() >= 0 and x.max() < 97
    assert all(len(set(r)) == 30 for r in x)
    cnt = np.bincount(x.reshape(-1), minlength=97)
    chi2 = ((cnt - cnt.mean()) ** 2 / cnt.mean()).sum()
    assert chi2 < 150          # 96 dof: p(chi2 > 150) ~ 3e-4
    full = ops.sample_items(50, 64, 64, False, 1, 1).cpu().numpy()      # S == n_items -> a permutation
    assert all(sorted(r) == list(range(64)) for r in full)
    y = ops.sample_items(4000, 97, 30, True, 123, 1).cpu().numpy()
    cnt = np.bincount(y.reshape(-1), minlength=97)
    assert ((cnt - cnt.mean()) ** 2 / cnt.mean()).sum() < 150
    a = ops.sample_items(10, 1000, 20, False, 1, 1).cpu().numpy()
    b = ops.sample_items(10, 1000, 20, False, 1, 2).cpu().numpy()
    assert not np.array_equal(a, b)          # a new step draws new samples
    with pytest.raises(RuntimeError):
        ops.sample_items(3, 5, 6, False, 0, 0)      # larger sample than population (np.random.choice raises too)


# ------------------------------------------------------------------------------------------------ K8
@pytest.mark.parametrize("n", [1, 3, 4, 1027, 100000])
def test_adam_tf_bit_exact(ops, n):
    rng = np.random.default_rng(n)
    w = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    wt, mt, vt = dev(w.copy()), dev(m.copy()), dev(v.copy())
    l2 = np.float32(3e-4)
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        lr_t = O.adam_lr_t(0.1, t)
        ops.adam_tf_step(wt, mt, vt, dev(g), float(lr_t), float(l2))
        O.adam_tf_step(w, m, v, g + w * l2, lr_t)
        assert np.array_equal(wt.cpu().numpy(), w), "step %d" % t
        assert np.array_equal(mt.cpu().numpy(), m) and np.array_equal(vt.cpu().numpy(), v)


# ------------------------------------------------------------------------------------------------ goldens
def test_goldens_on_gpu_recommendation_graphs(goldens):
    from tensorrec_amd import recommendation_graphs as R
    from tensorrec_amd.prediction_graphs import CosineSimilarityPredictionGraph
    g = goldens["bias_prediction_dense"]
    got = R.bias_prediction_dense(dev(g["predictions"]), dev(g["projected_user_biases"].astype(np.float32)),
                                  dev(g["projected_item_biases"].astype(np.float32))).cpu().numpy()
    assert (got == g["expected_biased_predictions"]).all()
    g = goldens["bias_prediction_serial"]
    got = R.bias_prediction_serial(dev(g["predictions"]), dev(g["projected_user_biases"].astype(np.float32)),
                                   dev(g["projected_item_biases"].astype(np.float32)),
                                   dev(g["x_user"].astype(np.int64)), dev(g["x_item"].astype(np.int64))).cpu().numpy()
    assert (got == g["expected_biased_predictions"]).all()
    g = goldens["densify_sampled_item_predictions"]
    got = R.densify_sampled_item_predictions(dev(g["input_data"]), 4, 3).cpu().numpy()
    assert (got == g["expected_result"]).all()
    g = goldens["collapse_mixture_of_tastes"]
    got = R.collapse_mixture_of_tastes([dev(p) for p in g["predictions"]], None).cpu().numpy()
    assert (got == g["expected_predictions"]).all()
    g = goldens["collapse_mixture_of_tastes_with_attention"]
    got = R.collapse_mixture_of_tastes([dev(p) for p in g["predictions"]], [dev(a) for a in g["attentions"]])
    assert np.allclose(got.cpu().numpy(), g["expected_predictions"], rtol=3e-7, atol=0)
    g = goldens["predict_similar_items"]
    got = R.predict_similar_items(CosineSimilarityPredictionGraph(), dev(g["reprs"]), [1]).cpu().numpy()
    assert (got == g["expected_sims"]).all()


# ------------------------------------------------------------------------------------------------ item shards
@pytest.mark.parametrize("world", [2, 8])
def test_item_shards_merge_to_global_topk(ops, world):
    """What N ranks do, on one GPU: per-shard fused top-k with global ids (item_index_base), lists concatenated the
    way the all-gather lays them out, HIP merge -> the exact global top-k (tests/test_sharding_gloo.py covers the
    collective itself)."""
    from tensorrec_amd import sharding
    u, v = _uv(130, 5000, 64, seed=world, integer=True)
    rng = np.random.default_rng(2)
    ub, ib = np.round(rng.standard_normal(130)).astype(np.float32), np.round(rng.standard_normal(5000)).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    lists_v, lists_i = [], []
    for r in range(world):
        b, e = sharding.shard_bounds(5000, world, r, align=64)
        v_op, _, _ = ops.score_prep(dev(v[b:e]), ops.DTYPE_F32)
        lv, li = ops.score_topk(u_op, v_op, ops.DTYPE_F32, kpad, 10, dev(ub), dev(ib[b:e]), item_index_base=b)
        lists_v.append(lv)
        lists_i.append(li)
    gv, gi = sharding.merge_topk(torch.cat(lists_v, dim=1), torch.cat(lists_i, dim=1), 10)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), 10)
    assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gv.cpu().numpy(), rv)


@pytest.mark.parametrize("world,n_items,dtype_name", [(2, 40000, "f32"), (8, 70000, "bf16"), (4, 3000, "f32")])
def test_item_shards_two_stage_with_shared_floor(ops, world, n_items, dtype_name):
    """Item shards + two-stage top-k + the shared floor (what bench.py --gpus N runs): the k-th largest superblock
    maximum over ALL shards bounds the global k-th best score from below (a first pass collects what every rank would
    all-gather, a second pass injects the floor where the collective would sit), superblocks below it are not
    re-scored, and the merged lists are still the exact global top-k, ties included (integer-valued scores make ties
    abundant).  The last case has fewer superblocks than k per shard."""
    from tensorrec_amd import sharding
    dt = ops.DTYPE_F32 if dtype_name == "f32" else ops.DTYPE_BF16
    n_users, k = 200, 10
    u, v = _uv(n_users, n_items, 128, seed=world, integer=(dtype_name == "f32"))
    rng = np.random.default_rng(2)
    ub = np.round(rng.standard_normal(n_users)).astype(np.float32)
    ib = np.round(rng.standard_normal(n_items)).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), dt)
    shards = []
    for r in range(world):
        b, e = sharding.shard_bounds(n_items, world, r, align=64)
        v_op, _, _ = ops.score_prep(dev(v[b:e]), dt)
        shards.append((b, v_op, dev(ib[b:e])))
    maxima = []

    def grab(sel_max):                      # pass 1: what every rank would contribute to the all-gather
        maxima.append(sel_max.clone())
        return torch.full((n_users,), float('-inf'), device=sel_max.device)

    for b, v_op, ibs in shards:
        ops.score_topk_two_stage(u_op, v_op, dt, kpad, k, dev(ub), ibs, item_index_base=b, floor_exchange=grab)
    floor = sharding.kth_largest_block_max(torch.cat(maxima, dim=0), k)
    lists_v, lists_i, kept = [], [], 0
    for b, v_op, ibs in shards:
        lv, li = ops.score_topk_two_stage(u_op, v_op, dt, kpad, k, dev(ub), ibs, item_index_base=b,
                                          floor_exchange=lambda sel_max: floor)
        lists_v.append(lv)
        lists_i.append(li)
        kept += int((li >= 0).sum())
    gv, gi = sharding.merge_topk(torch.cat(lists_v, dim=1), torch.cat(lists_i, dim=1), k)
    if dtype_name == "f32":
        scores = O.score_dense_exact(u, v, ub, ib)
    else:
        v_full, _, _ = ops.score_prep(dev(v), dt)
        scores = ops.score_store(u_op, v_full, dt, kpad, dev(ub), dev(ib)).cpu().numpy()
    rv, ri = O.topk_rows(scores, k)
    assert np.array_equal(gi.cpu().numpy(), ri) and np.array_equal(gv.cpu().numpy(), rv)
    assert (floor.cpu().numpy() <= rv[:, k - 1]).all()       # a lower bound of the true k-th best score
    # the floor did prune: without it every shard returns k items per user (continuous scores: ties are rare)
    if dtype_name == "bf16":
        assert kept < 0.5 * world * n_users * k


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
def test_score_topk_bf16_all_tilings(ops, variant):
    """Every tiling of the hot configuration (bf16, K=128, top-10 -> capacity 12) gives the exact top-k of the bf16
    score matrix, with biases, ragged sizes and several item chunks."""
    u, v = _uv(777, 4099, 128, seed=11)
    rng = np.random.default_rng(3)
    ub, ib = rng.standard_normal(777).astype(np.float32), rng.standard_normal(4099).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    scores = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, dev(ub), dev(ib)).cpu().numpy()
    for chunks in (1, 3):
        vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_BF16, kpad, 10, dev(ub), dev(ib), n_chunks=chunks,
                                   variant=variant)
        rv, ri = O.topk_rows(scores, 10)
        assert np.array_equal(vals.cpu().numpy(), rv), "chunks=%d" % chunks
        assert np.array_equal(idx.cpu().numpy(), ri), "chunks=%d" % chunks


@pytest.mark.parametrize("n_users,n_items,sb_rows", [(5, 70, 128), (513, 129, 128), (700, 1025, 256), (64, 64, 512)])
@pytest.mark.parametrize("kdim", [128, 64])
def test_blockmax_kernel_small_and_ragged_sizes(ops, n_users, n_items, sb_rows, kdim):
    """The hand-scheduled stage-1 kernel on sizes around its tile / workgroup / superblock boundaries (one tile, one
    partial tile, a single workgroup with mostly idle rows, superblocks of 2, 4 and 8 tiles), K = 128 and K = 64."""
    u, v = _uv(n_users, n_items, kdim, seed=n_users)
    rng = np.random.default_rng(1)
    ub, ib = rng.standard_normal(n_users).astype(np.float32), rng.standard_normal(n_items).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    scores = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, dev(ub), dev(ib)).cpu().numpy()
    for k in (1, 10):
        vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_BF16, kpad, k, dev(ub), dev(ib), sb_rows=sb_rows)
        rv, ri = O.topk_rows(scores, k)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv), k


@pytest.mark.parametrize("shape", [2, 3, 4, 5])
@pytest.mark.parametrize("biased", [True, False])
def test_blockmax_kernel_shapes_are_exact(ops, shape, biased):
    """Every register plan of the hand-scheduled stage-1 kernel (users per wave / accumulator sets / workgroups per CU)
    and the generic kernel produce the same superblock maxima, hence the exact top-k of the bf16 score matrix -- ragged
    user and item counts, several chunks, superblocks that end inside a tile."""
    import tensorrec_amd as T
    u, v = _uv(1000, 9000 + 37, 128, seed=shape)
    rng = np.random.default_rng(3)
    ub = rng.standard_normal(u.shape[0]).astype(np.float32) if biased else None
    ib = rng.standard_normal(v.shape[0]).astype(np.float32) if biased else None
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    dub, dib = (dev(ub), dev(ib)) if biased else (None, None)
    scores = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, dub, dib).cpu().numpy()
    rv, ri = O.topk_rows(scores, 10)
    T._native.set_tuning("blockmax_shape", shape)
    try:
        for chunks in (None, 3):
            vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_BF16, kpad, 10, dub, dib, n_chunks=chunks)
            assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv), (shape, chunks)
    finally:
        T._native.set_tuning("blockmax_shape", 5)
    T._native.set_tuning("blockmax_pipelined", 0)
    try:
        vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_BF16, kpad, 10, dub, dib)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
    finally:
        T._native.set_tuning("blockmax_pipelined", 1)


def test_group_pairs_by_item(ops):
    """device counting sort: every pair lands exactly once in its item's bucket; buckets follow item order."""
    rng = np.random.default_rng(0)
    for (nu, ni, S) in ((7, 5, 3), (300, 2500, 40), (1000, 97, 30)):
        items = rng.integers(0, ni, (nu, S)).astype(np.int32)
        xi = dev(items.reshape(-1))
        indptr_t, users_t, perm_t = ops.group_pairs_by_item(None, xi, S, ni)
        indptr_t, users_t, perm_t = indptr_t.cpu().numpy(), users_t.cpu().numpy(), perm_t.cpu().numpy()
        counts = np.bincount(items.reshape(-1), minlength=ni)
        assert np.array_equal(np.diff(indptr_t), counts) and indptr_t[0] == 0 and indptr_t[-1] == nu * S
        assert sorted(perm_t.tolist()) == list(range(nu * S))
        assert np.array_equal(users_t, perm_t // S)
        bucket_of_slot = np.repeat(np.arange(ni), counts)
        assert np.array_equal(items.reshape(-1)[perm_t], bucket_of_slot)
    # few buckets + many pairs: the privatised (LDS-counter) path; negative keys belong to no bucket
    nu, ni, S = 5000, 300, 16
    items = rng.integers(-1, ni, (nu, S)).astype(np.int32)
    flat = items.reshape(-1)
    indptr_t, users_t, perm_t = [t.cpu().numpy() for t in ops.group_pairs_by_item(None, dev(flat), S, ni)]
    kept = flat >= 0
    counts = np.bincount(flat[kept], minlength=ni)
    n_kept = int(kept.sum())
    assert np.array_equal(np.diff(indptr_t), counts) and indptr_t[-1] == n_kept
    assert sorted(perm_t[:n_kept].tolist()) == np.nonzero(kept)[0].tolist()
    assert np.array_equal(users_t[:n_kept], perm_t[:n_kept] // S)
    assert np.array_equal(flat[perm_t[:n_kept]], np.repeat(np.arange(ni), counts))
    # explicit user ids
    xu = rng.integers(0, 50, 999).astype(np.int32)
    xi = rng.integers(0, 70, 999).astype(np.int32)
    indptr_t, users_t, perm_t = ops.group_pairs_by_item(dev(xu), dev(xi), 0, 70)
    assert np.array_equal(users_t.cpu().numpy(), xu[perm_t.cpu().numpy()])


# ------------------------------------------------------------------------------------------------ shim goldens
def test_hip_path_vs_reference_source_fixtures(ops):
    """HIP kernels against fixtures produced by executing the reference's own graph source on the NumPy TF stand-in
    (tests/golden/run_reference_on_shim.py): representation graphs, losses, prediction graphs, ranks."""
    from conftest import load_goldens
    from tensorrec_amd.sparse import Interactions
    sg = load_goldens("reference_shim_goldens.json")
    tol = dict(rtol=1e-5, atol=1e-5)
    g = sg["repr_linear"]
    f = feats(g["features"])
    w = dev(g["variables"]["linear_weights_user"])
    assert np.allclose(ops.sparse_dense_matmul(f, w).cpu().numpy(), g["expected_repr"], **tol)
    g = sg["repr_normalized_linear"]
    got = ops.sparse_dense_matmul_l2norm(feats(g["features"]), dev(g["variables"]["linear_weights_user"]))
    assert np.allclose(got.cpu().numpy(), g["expected_repr"], **tol)
    for key in ("repr_relu", "repr_relu_size_5"):
        g = sg[key]
        v = g["variables"]
        h = ops.sparse_dense_matmul_bias_relu(feats(g["features"]), dev(v["relu_weights_user"]), dev(v["relu_biases_user"]))
        got = ops.matmul(h, dev(v["linear_weights_user"]))
        assert np.allclose(got.cpu().numpy(), g["expected_repr"], rtol=1e-5, atol=1e-5)
    g = sg["repr_passthrough"]
    assert np.array_equal(ops.sparse_to_dense(feats(g["features"])).cpu().numpy(), g["expected_repr"])
    # losses
    g = sg["loss_wmrb"]
    m = sp.csr_matrix(g["interactions"])
    inter = Interactions(m, m.shape[0], m.shape[1], "cuda")
    for key, balanced in (("loss_wmrb", False), ("loss_balanced_wmrb", True)):
        g = sg[key]
        got = ops.wmrb_loss(dev(g["prediction_serial"]), dev(g["sample_predictions"]), inter, balanced=balanced)
        assert np.allclose(got.cpu().numpy(), g["expected_loss"], **tol)
    g = sg["loss_rmse"]
    assert np.allclose(float(ops.rmse_loss(dev(g["prediction_serial"]), inter.values)), g["expected_loss"], rtol=1e-5)
    # dense + separation loss graphs (loss_graphs.py:62-134) through the product classes
    from tensorrec_amd import loss_graphs as LG
    from tensorrec_amd.prediction_graphs import DotProductPredictionGraph as Dot
    g = sg["loss_separation"]
    got = LG.SeparationLossGraph().connect_loss_graph(tf_prediction_serial=dev(g["prediction_serial"]),
                                                      tf_interactions_serial=inter.values)
    assert np.allclose(float(got), g["expected_loss"], rtol=1e-5, atol=1e-6)
    for key, cls in (("loss_rmse_dense", LG.RMSEDenseLossGraph), ("loss_separation_dense", LG.SeparationDenseLossGraph)):
        g = sg[key]
        dense = Dot().connect_dense_prediction_graph(dev(g["user_repr"]), dev(g["item_repr"]))
        got = cls().connect_loss_graph(tf_prediction=dense, tf_interactions=inter)
        assert np.allclose(float(got), g["expected_loss"], rtol=1e-5, atol=1e-6), key
    # prediction graphs
    from tensorrec_amd.prediction_graphs import (DotProductPredictionGraph, CosineSimilarityPredictionGraph,
                                                 EuclideanSimilarityPredictionGraph)
    for kind, cls in (("dot", DotProductPredictionGraph), ("cosine", CosineSimilarityPredictionGraph),
                      ("euclidean", EuclideanSimilarityPredictionGraph)):
        g = sg["pred_" + kind]
        u, v = dev(g["user_repr"]), dev(g["item_repr"])
        assert np.allclose(cls().connect_dense_prediction_graph(u, v).cpu().numpy(), g["expected_dense"], **tol)
        xu, xi = dev(g["x_user"].astype(np.int64)), dev(g["x_item"].astype(np.int64))
        assert np.allclose(cls().connect_serial_prediction_graph(u, v, xu, xi).cpu().numpy(), g["expected_serial"], **tol)
    g = sg["rank_predictions_ties"]
    assert np.array_equal(ops.rank_rows(dev(g["predictions"])).cpu().numpy(), g["expected_ranks"].astype(np.int32))


# ------------------------------------------------------------------------------------------------ two-stage top-k
@pytest.mark.parametrize("sb_rows,chunks", [(128, None), (256, 1), (512, 3)])
@pytest.mark.parametrize("integer", [False, True])
def test_two_stage_topk_fp32_exact(ops, sb_rows, chunks, integer):
    """superblock maxima -> select -> grouped re-score -> merge == the reference's top-k bit for bit, ties included
    (integer operands: thousands of exact ties, many superblocks with equal maxima)."""
    u, v = _uv(333, 5000, 64, seed=sb_rows, integer=integer)
    rng = np.random.default_rng(4)
    ub, ib = rng.standard_normal(333).astype(np.float32), rng.standard_normal(5000).astype(np.float32)
    if integer:
        ub, ib = np.round(ub), np.round(ib)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    ref = O.score_dense_exact(u, v, ub, ib)
    for k in (1, 10, 16):
        vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_F32, kpad, k, dev(ub), dev(ib), sb_rows=sb_rows,
                                             n_chunks=chunks, item_index_base=0)
        rv, ri = O.topk_rows(ref, k)
        assert np.array_equal(idx.cpu().numpy(), ri), "k=%d" % k
        assert np.array_equal(vals.cpu().numpy(), rv), "k=%d" % k
    # unbiased + item shard offset
    vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_F32, kpad, 10, sb_rows=sb_rows, item_index_base=700)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v), 10)
    assert np.array_equal(idx.cpu().numpy(), ri + 700) and np.array_equal(vals.cpu().numpy(), rv)


def test_two_stage_topk_fewer_superblocks_than_k(ops):
    u, v = _uv(100, 300, 32, seed=1)          # 3 superblocks of 128 < k = 10
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_F32)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_F32)
    vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_F32, kpad, 10, sb_rows=128)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v), 10)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)


@pytest.mark.parametrize("variant", [0, 1])
def test_two_stage_topk_bf16_and_euclid(ops, variant):
    u, v = _uv(700, 20000, 128, seed=9)
    rng = np.random.default_rng(5)
    ub, ib = rng.standard_normal(700).astype(np.float32), rng.standard_normal(20000).astype(np.float32)
    u_op, _, kpad = ops.score_prep(dev(u), ops.DTYPE_BF16)
    v_op, _, _ = ops.score_prep(dev(v), ops.DTYPE_BF16)
    scores = ops.score_store(u_op, v_op, ops.DTYPE_BF16, kpad, dev(ub), dev(ib)).cpu().numpy()
    vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_BF16, kpad, 10, dev(ub), dev(ib), variant=variant)   # auto -> two-stage
    rv, ri = O.topk_rows(scores, 10)
    assert np.array_equal(vals.cpu().numpy(), rv) and np.array_equal(idx.cpu().numpy(), ri)
    dv, di = ops.score_topk(u_op, v_op, ops.DTYPE_BF16, kpad, 10, dev(ub), dev(ib), variant=variant, method="direct")
    assert np.array_equal(dv.cpu().numpy(), rv) and np.array_equal(di.cpu().numpy(), ri)
    # euclidean, fp32
    u, v = _uv(130, 17000, 40, seed=3)
    u_op, u_sq, kpad = ops.score_prep(dev(u), ops.DTYPE_F32, want_sqnorm=True)
    v_op, v_sq, _ = ops.score_prep(dev(v), ops.DTYPE_F32, want_sqnorm=True)
    vals, idx = ops.score_topk(u_op, v_op, ops.DTYPE_F32, kpad, 5, mode=ops.MODE_EUCLIDEAN, user_sq=u_sq, item_sq=v_sq)
    ref = O.score_dense_euclid_exact(u, v, u_sq.cpu().numpy(), v_sq.cpu().numpy())
    rv, ri = O.topk_rows(ref, 5)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
