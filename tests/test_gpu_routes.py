"""Which route predict_top_k takes for every BASELINE.json configuration (and for the shapes the other routes exist for): the call
reports it (``model.last_route`` / ``return_route=True``) and these tests pin it -- a shape that silently falls onto a slower route
is the likeliest regression of that method.  Every call is also checked against the oracle's top-k on a few
users, so a pinned route that computes the wrong thing cannot pass either."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O

pytestmark = pytest.mark.gpu

import tensorrec_amd as T  # noqa: E402
from tensorrec_amd.prediction_graphs import (DotProductPredictionGraph, CosineSimilarityPredictionGraph,  # noqa: E402
                                             EuclideanSimilarityPredictionGraph)
from tensorrec_amd.representation_graphs import LinearRepresentationGraph, ReLURepresentationGraph  # noqa: E402


def _model(n_users, n_items, d, graph, precision="fp32", repr_graph=LinearRepresentationGraph, **kw):
    m = T.TensorRec(n_components=d, prediction_graph=graph(), user_repr_graph=repr_graph(), item_repr_graph=repr_graph(),
                    seed=0, precision=precision, **kw)
    m.build(n_users, n_items)
    w = m.get_weights()
    rng = np.random.default_rng(1)
    for name in ("user_feature_biases", "item_feature_biases"):
        w[name] = (0.05 * rng.standard_normal(w[name].shape)).astype(np.float32)
    m.set_weights(w)
    return m


def _check(model, uf, itf, k, route, exact=True, **kw):
    vals, idx, rep = model.predict_top_k(uf, itf, k=k, return_route=True, **kw)
    assert rep["route"] == route and rep == model.last_route, rep
    assert vals.shape == (uf.shape[0], k) and idx.shape == (uf.shape[0], k)
    if exact:                                                 # the first users against the dense prediction's top-k (oracle order)
        n = min(8, uf.shape[0])
        pred = model.predict(uf[:n], itf)
        rv, ri = O.topk_rows(pred, k)
        assert np.array_equal(idx[:n], ri) and np.array_equal(vals[:n], rv)
    return rep


def test_route_configs_0_and_1_direct():
    """configs[0] (README example, 100 x 150, d = 100) and configs[1] (943 x 1,682, d = 64): one fused pass with per-lane lists"""
    for (nu, ni, d) in ((100, 150, 100), (943, 1682, 64)):
        m = _model(nu, ni, d, DotProductPredictionGraph)
        _check(m, sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr"), 10, "direct")


def test_route_config_2_cascade_int8():
    """configs[2]: 1M items, d = 128, DotProduct, top-10 -> the int8 -> bf16 -> fp32 cascade (user count does not enter the choice)"""
    nu, ni = 2048, 1_000_000
    m = _model(nu, ni, 128, DotProductPredictionGraph)
    rep = _check(m, sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr"), 10, "cascade_int8")
    assert not rep["sharded"] and rep["user_batch_size"] >= nu


def test_route_config_3_shard_cascade_int8():
    """configs[3]: one rank's shard of the 10M-item catalogue (1.25M items), d = 128, CosineSimilarity, top-10"""
    nu, ni = 1024, 1_250_000
    m = _model(nu, ni, 128, CosineSimilarityPredictionGraph)
    _check(m, sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr"), 10, "cascade_int8")


def test_route_config_4_euclid():
    """configs[4]: 26,744 items, ReLU d = 256, Euclidean -> the certified Euclidean route (k <= 12)"""
    nu, ni = 512, 26_744
    m = _model(nu, ni, 256, EuclideanSimilarityPredictionGraph, repr_graph=ReLURepresentationGraph)
    _check(m, sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr"), 10, "euclid_certified")


def test_routes_off_the_headline_shapes():
    """17 <= k <= 64 on a cascade-sized catalogue -> wide_cascade; bf16 precision -> two_stage (approximate scores: not compared);
    a mid-sized catalogue (no int8 stage below 262,144 items) -> bf16_filter; attention models -> slab"""
    nu, ni = 512, 300_000
    uf, itf = sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr")
    m = _model(nu, ni, 128, DotProductPredictionGraph)
    _check(m, uf, itf, 32, "wide_cascade")
    _check(m, uf, itf, 10, "cascade_int8")
    mb = _model(nu, ni, 128, DotProductPredictionGraph, precision="bf16")
    _check(mb, uf, itf, 10, "two_stage", exact=False)
    nm = 100_000
    mm = _model(nu, nm, 64, DotProductPredictionGraph)
    _check(mm, uf, sp.identity(nm, dtype=np.float32, format="csr"), 10, "bf16_filter")
    ma = _model(64, 3000, 16, DotProductPredictionGraph, n_tastes=2, attention_graph=LinearRepresentationGraph())
    _check(ma, sp.identity(64, dtype=np.float32, format="csr"), sp.identity(3000, dtype=np.float32, format="csr"), 5, "slab")


def test_k_beyond_the_fused_lists_off_the_wide_route_takes_score_slabs():
    """k > 16 where the wide cascade does not run -- a small catalogue (configs[1]'s shape), Euclidean scores, bf16 precision, k > 64
    -- used to end in the fused kernels' "k <= 16" error: exact score slabs + the k best of every row answer instead, any k."""
    nu, ni = 300, 1682
    uf, itf = sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr")
    _check(_model(nu, ni, 64, DotProductPredictionGraph), uf, itf, 20, "slab")
    _check(_model(nu, ni, 64, EuclideanSimilarityPredictionGraph), uf, itf, 40, "slab")
    _check(_model(nu, ni, 64, CosineSimilarityPredictionGraph, n_tastes=2), uf, itf, 17, "slab")
    big = 300_000
    m = _model(nu, big, 128, DotProductPredictionGraph)
    itb = sp.identity(big, dtype=np.float32, format="csr")
    _check(m, uf, itb, 100, "slab")                           # (above the wide route's 64)
    _check(m, uf, itb, 64, "wide_cascade")
    me = _model(nu, big, 128, EuclideanSimilarityPredictionGraph)
    _check(me, uf, itb, 20, "euclid_certified")               # (13 <= k <= 48 where the int8 cascade runs: the wide lists, certified)
    _check(me, uf, itb, 60, "slab")
    with pytest.raises(ValueError):
        m.predict_top_k(uf, itb, k=0)


@pytest.mark.parametrize("nu,ni,k", [(7, 5, 10), (7, 5, 5), (7, 30, 40), (3, 1, 1)])
def test_k_at_and_beyond_the_catalogue_size(nu, ni, k):
    """k >= n_items on either kind of route: the first n_items places are the oracle's order, the places beyond hold -inf / -1."""
    m = _model(nu, ni, 8, DotProductPredictionGraph)
    uf, itf = sp.identity(nu, dtype=np.float32, format="csr"), sp.identity(ni, dtype=np.float32, format="csr")
    vals, idx = m.predict_top_k(uf, itf, k=k)
    kk = min(k, ni)
    rv, ri = O.topk_rows(m.predict(uf, itf), kk)
    assert np.array_equal(idx[:, :kk], ri) and np.array_equal(vals[:, :kk], rv)
    assert np.all(idx[:, kk:] == -1) and np.all(np.isneginf(vals[:, kk:]))
