"""The argument behind the cascade's candidate lists (header of trec_score_gemm_refine_candidates in
csrc/topk_cascade.hip), replayed in NumPy on the oracle's fp32 scores -- no GPU: for ANY lower bound tauLB <= t_k (the true k-th
best score) and any bf16-path score sh with |sh - s| <= eps, the items listed with the provisional floor tauLB - eps hold
the exact top-k, and the survivors of floor = (k-th largest listed sh) - 2 eps still do.  eps is the bound of
csrc/topk_filter.hip (filter_eps) evaluated as the kernels evaluate it; sh is emulated in three ways: operands rounded to
bf16 and contracted in float32 (what the MFMA path computes, up to summation order), and the adversarial extremes
sh = s +/- eps per item.  Replaces, as a filter, tf.matmul + tf.nn.top_k of tensorrec/prediction_graphs.py:49-50 and
tensorrec/recommendation_graphs.py:80."""
import numpy as np
import pytest

from oracle import oracle as O


def bf16_round(x):
    """float32 -> nearest-even bfloat16 -> float32 (v_cvt_pk_bf16_f32)."""
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def filter_eps(x, xh, items, items_h, bu, bi_max, kdim):
    """eps_u of csrc/topk_filter.hip: ||x - xh|| max ||yh|| + ||x|| max ||y - yh|| + (K + 2)(2^-24 + 2^-22)(||x|| max ||y|| + |b_u| + max |b_i|),
    inflated as the kernel inflates it."""
    nx = np.linalg.norm(x, axis=1)
    ex = np.linalg.norm(x - xh, axis=1)
    ni = np.linalg.norm(items, axis=1).max()
    ai = np.linalg.norm(items - items_h, axis=1).max()
    ck = (kdim + 2) * 2.98023224e-07
    eps = ex * (ni * 1.00390625) + nx * ai + ck * (nx * ni * 1.0078125 + np.abs(bu) + bi_max)
    return (eps * 1.001953125 + 1e-30).astype(np.float32)


@pytest.mark.parametrize("seed,kind", [(0, "gauss"), (1, "clustered"), (2, "scaled"), (3, "ties")])
@pytest.mark.parametrize("k", [1, 10, 16])
def test_lists_made_with_tauLB_minus_eps_hold_the_top_k_and_survive_the_floor(seed, kind, k):
    rng = np.random.default_rng(seed)
    n_u, n_i, d = 40, 6000, 64
    u = rng.standard_normal((n_u, d)).astype(np.float32)
    v = rng.standard_normal((n_i, d)).astype(np.float32)
    if kind == "clustered":
        v = (rng.standard_normal((12, d))[rng.integers(0, 12, n_i)] + 0.02 * v).astype(np.float32)
    if kind == "scaled":
        u = (u * np.exp(2.0 * rng.standard_normal((n_u, 1)))).astype(np.float32)
    if kind == "ties":
        v[100:140] = v[7]                                   # forty copies of one item: exact ties, index order decides
    ub = (0.3 * rng.standard_normal(n_u)).astype(np.float32)
    ib = (0.3 * rng.standard_normal(n_i)).astype(np.float32)
    s = O.score_dense_exact(u, v, ub, ib)                   # the reference's fp32 scores
    tv, ti = O.topk_rows(s, k)
    uh, vh = bf16_round(u), bf16_round(v)
    eps = filter_eps(u, uh, v, vh, ub, np.abs(ib).max(), d)
    sh_mfma = ((uh.astype(np.float64) @ vh.astype(np.float64).T).astype(np.float32) + ub[:, None]) + ib[None, :]
    assert (np.abs(sh_mfma - s) <= eps[:, None]).all()      # the bound itself, on this data
    sign = np.where(rng.random(s.shape) < 0.5, -1.0, 1.0).astype(np.float32)
    for sh in (sh_mfma, s + sign * eps[:, None] * 0.999, s - eps[:, None] * 0.999 * (s >= tv[:, -1:]) + eps[:, None] * 0.999 * (s < tv[:, -1:])):
        for slack in (0.0, 0.5, 3.0):                       # tauLB = t_k - slack * eps: any lower bound of t_k will do
            tau_lb = tv[:, -1] - slack * eps
            f0 = tau_lb - eps
            for r in range(n_u):
                listed = np.flatnonzero(sh[r] >= f0[r])
                assert np.isin(ti[r], listed).all()                         # a top-k item is listed
                tau = np.sort(sh[r, listed])[-k]                            # (at least k are)
                survivors = listed[sh[r, listed] >= tau - 2.0 * eps[r]]
                assert np.isin(ti[r], survivors).all()                      # ... and survives the finish kernel's floor
                # the exact order among the survivors is the exact order of the first k places
                order = survivors[np.lexsort((survivors, -s[r, survivors]))][:k]
                assert np.array_equal(order, ti[r])
