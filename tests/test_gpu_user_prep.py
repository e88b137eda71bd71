"""The user side of the cascade prepared on the device (csrc/user_prep.hip, ops.score_prep_filter(sort_users=True)): the layout
is a stable sort by scale class with bands of two classes padded to whole int8 workgroups, and the operands written by the one
gather pass equal what the stand-alone preparation kernels make of the same rows.  No host read on the path: the layout's row
count is a bound (trec_user_prep_alloc_rows)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tensorrec_amd import ops as _ops
    return _ops


def _rows(kind, n, d, rng):
    x = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "scaled":                       # row scales over six orders of magnitude: many classes, many bands
        x *= np.exp(rng.uniform(-7, 7, size=(n, 1))).astype(np.float32)
    elif kind == "outliers":                   # a few huge elements: the row wants to clip them
        x[rng.random((n, d)) < 0.01] *= 60.0
    elif kind == "zeros":
        x[::5] = 0.0
    return x


@pytest.mark.parametrize("kind,n,d,k,normalize", [("gauss", 5000, 128, 10, False), ("scaled", 7001, 128, 10, False),
                                                  ("outliers", 3000, 64, 16, False), ("zeros", 2500, 100, 10, False),
                                                  ("scaled", 4000, 128, 10, True), ("gauss", 300, 40, 10, False)])
def test_layout_is_a_stable_class_sort_with_padded_bands(ops, kind, n, d, k, normalize):
    from tensorrec_amd import _native as N
    rng = np.random.default_rng(n + d)
    x = _rows(kind, n, d, rng)
    ub = rng.standard_normal(n).astype(np.float32)
    dx, dub = torch.from_numpy(x).cuda(), torch.from_numpy(ub).cuda()
    op = ops.score_prep_filter(dx, normalize=normalize, sort_users=True, k=k, user_bias=dub)
    wg = int(N.query("trec_score_blockmax_i8_rows_per_workgroup", 10 if k <= 10 else 16))
    assert op.wg_rows == wg and op.n == int(N.query("trec_user_prep_alloc_rows", n, wg)) and op.n % wg == 0 and op.n_real == n
    src, pos = op.src.cpu().numpy(), op.pos.cpu().numpy()
    n_pad, n_real = op.meta.cpu().numpy()
    assert n_real == n and n_pad % wg == 0 and n <= n_pad <= op.n
    # ---- a permutation: every user exactly once, pos is its inverse, nothing beyond the padded rows
    real = np.flatnonzero(src >= 0)
    assert len(real) == n and np.array_equal(np.sort(src[real]), np.arange(n))
    assert np.array_equal(pos[src[real]], real) and real.max() < n_pad
    # ---- classes from the scales the kernel itself reports (ladder), recomputed here from the operand rows
    ladder = op.ladder.cpu().numpy()
    gmax = float(op.gmax.cpu().numpy()[0])
    assert np.allclose(ladder, gmax * 2.0 ** (-np.arange(64) / 4.0), rtol=1e-6)
    wg_class, wg_scale = op.wg_class.cpu().numpy(), op.wg_scale.cpu().numpy()
    n_wg = op.n // wg
    assert (wg_class[n_pad // wg:] == -1).all() and (wg_scale[n_pad // wg:] == 0).all()
    assert (wg_class[:n_pad // wg] >= 0).all() and np.allclose(wg_scale[:n_pad // wg], ladder[wg_class[:n_pad // wg]])
    used = op.class_used.cpu().numpy()
    assert set(np.flatnonzero(used)) == set(wg_class[:n_pad // wg].tolist())
    # the class of a row is not observable directly; what must hold: along the layout the real rows' classes never decrease,
    # users of one class keep their order (stable), a workgroup's rows come from ONE band of two classes and its class is the
    # smallest class it holds.  Recompute the classes like the kernel does (float32 arithmetic; ties at class edges may differ
    # by the last bit of log2f, so the check is on the ORDER the layout implies)
    nat_ref, decided = _wanted_scale(x, normalize)
    g = gmax if gmax > 0 else 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        cls = np.floor(4.0 * np.log2(np.float32(g) / nat_ref.astype(np.float32)))
    cls = np.nan_to_num(np.clip(cls, 0, 63), nan=0.0).astype(np.int64)
    c_layout = cls[src[real]]
    edge = np.abs(4.0 * np.log2(g / np.maximum(nat_ref[src[real]], 1e-38)) % 1.0)
    clear = (edge > 1e-4) & (edge < 1 - 1e-4) & decided[src[real]]  # rows whose class does not hinge on the last bit
    assert (np.diff(c_layout[clear]) >= 0).all()
    for c in np.unique(c_layout[clear]):
        rows_c = src[real][clear & (c_layout == c)]
        assert (np.diff(rows_c) > 0).all()                         # stable: caller order inside a class
    w_of = real // wg
    band = c_layout // 2
    for w in np.unique(w_of):
        m = (w_of == w) & clear
        if m.any():
            assert len(np.unique(band[m])) == 1 and wg_class[w] // 2 == band[m][0] and wg_class[w] <= c_layout[m].min()
    # ---- operands: the gather pass == the stand-alone kernels on the gathered rows
    perm = np.clip(src, 0, None)
    g_rows = torch.from_numpy(x[perm]).cuda()
    ref = ops.score_prep_filter(g_rows, normalize=normalize)
    keep = torch.from_numpy(src >= 0).cuda()
    assert torch.equal(op.bf16[keep].view(torch.int16), ref.bf16[keep].view(torch.int16))
    assert torch.equal(op.f32[keep], ref.f32[keep])
    assert torch.allclose(op.stats[keep], ref.stats[keep], rtol=2e-6, atol=0)
    q = torch.empty_like(op.i8)
    st8 = torch.empty_like(op.stats8)
    scale_rows = op.wg_scale.clone()
    scale_rows[scale_rows == 0] = 1.0
    N.call("trec_score_prep_i8_users", N.ptr(ref.f32), op.n, ref.f32.shape[1], op.kpad, N.ptr(scale_rows), wg, N.ptr(q), N.ptr(st8))
    assert torch.equal(op.i8[keep], q[keep])
    assert torch.allclose(op.stats8[keep], st8[keep], rtol=2e-6, atol=1e-30)
    # rows without a source are zero and carry zero statistics
    assert not op.i8[~keep].any() and not op.f32[~keep].any() and not op.stats[~keep].any() and not op.stats8[~keep].any()
    assert torch.equal(op.bias_sorted[keep], dub[torch.from_numpy(perm).cuda()][keep]) and not op.bias_sorted[~keep].any()


def _wanted_scale(x, normalize):
    """NumPy restatement of natscale_kernel: the best of max|x|/127 and a half / a quarter of it by quantisation error."""
    am = np.abs(x).max(axis=1)
    best = am / 127.0
    with np.errstate(divide="ignore", invalid="ignore"):
        errs = []
        for f in (1.0, 0.5, 0.25):
            sc = (am / 127.0 * f)[:, None]
            q = np.clip(np.rint(x / sc), -127, 127)
            errs.append(((x - q * sc) ** 2).sum(axis=1))
    e0, e1, e2 = errs
    half = (e1 < e0) & (e1 <= e2)
    quarter = ~half & (e2 < e0) & (e2 < e1)
    best = np.where(half, best * 0.5, np.where(quarter, best * 0.25, best))
    best = np.where(am > 0, best, 0.0)
    if normalize:
        best = best / np.maximum(np.sqrt((x.astype(np.float64) ** 2).sum(axis=1)), 1e-6)
    # rows whose choice between the three candidates is not a near-tie (the kernel sums the errors in float32, in lane order)
    rel = lambda a, b: np.abs(a - b) > 1e-3 * np.maximum(np.maximum(a, b), 1e-30)          # noqa: E731
    decided = rel(e0, e1) & rel(e0, e2) & rel(e1, e2) | (am == 0)
    return best, decided


def test_sorted_operand_gives_the_oracle_topk_with_and_without_the_bias_in_the_prep(ops):
    """End to end through the cascade: results leave in the caller's order (the finish writes through src), whether the user
    bias travelled through the preparation or is permuted by the call."""
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    n_u, n_i, d, k = 2100, 300_000, 128, 10
    u = (rng.standard_normal((n_u, d)) * np.exp(rng.uniform(-2, 2, size=(n_u, 1)))).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    ub = (0.3 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.3 * rng.standard_normal(n_i)).astype(np.float32)
    du, dv, dub, dib = (torch.from_numpy(a).cuda() for a in (u, v, ub, ib))
    iop = ops.score_prep_filter(dv, bias=dib, want_gstats=True)
    rv, ri = O.topk_rows(O.score_dense_exact(u, v, ub, ib), k)
    for with_bias in (True, False):
        uop = ops.score_prep_filter(du, sort_users=True, k=k, user_bias=dub if with_bias else None)
        vals, idx = ops.score_topk_filtered(uop, iop, k, dub, dib, prefilter="int8")
        assert ops.LAST_FILTER_STATS.get("prefilter") == "int8" and ops.LAST_FILTER_STATS.get("tail") == "candidate lists"
        assert vals.shape == (n_u, k)
        assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy(), rv)
