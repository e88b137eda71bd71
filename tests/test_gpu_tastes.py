"""Mixture of tastes / attention: the K9 collapse kernel against the NumPy oracle and torch-CPU
autograd, and mixture-of-tastes models against the oracle model (tensorrec.py:339-418 wiring, including the reference's
sample-attention quirk).  The API shape checks restate test/test_tensorrec.py:299-397 (n_tastes=3, attention)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import oracle as O
from oracle.model import OracleTensorRec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tensorrec_amd
    tensorrec_amd._native.require_gpu()
    return tensorrec_amd


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------ K9 kernel
@pytest.mark.parametrize("n_tastes", [1, 2, 3, 16])
@pytest.mark.parametrize("attention", [False, True])
def test_collapse_forward_vs_oracle(T, n_tastes, attention):
    rng = np.random.default_rng(n_tastes)
    n = 100003
    p = rng.standard_normal((n_tastes, n)).astype(np.float32)
    p[:, ::7] = np.round(p[:, ::7])                                # ties between tastes
    a = (3 * rng.standard_normal((n_tastes, n))).astype(np.float32) if attention else None
    got = T.ops.collapse_tastes(dev(p), dev(a) if attention else None).cpu().numpy()
    ref = O.collapse_mixture_of_tastes(list(p), list(a) if attention else None)
    if attention:
        assert np.allclose(got, ref, rtol=2e-6, atol=1e-6)          # expf differs from NumPy's exp in the last ulp
    else:
        assert np.array_equal(got, ref)


def test_collapse_fused_biases_serial_and_dense(T):
    rng = np.random.default_rng(0)
    n_users, n_items, S, Tn = 37, 53, 5, 3
    ub, ib = rng.standard_normal(n_users).astype(np.float32), rng.standard_normal(n_items).astype(np.float32)
    # dense
    p = rng.standard_normal((Tn, n_users, n_items)).astype(np.float32)
    got = T.ops.collapse_tastes(dev(p), None, dev(ub), dev(ib)).cpu().numpy()
    assert np.array_equal(got, (p.max(0) + ub[:, None]) + ib[None, :])
    # serial with explicit users
    n = 1000
    xu, xi = rng.integers(0, n_users, n), rng.integers(0, n_items, n)
    ps = rng.standard_normal((Tn, n)).astype(np.float32)
    got = T.ops.collapse_tastes(dev(ps), None, dev(ub), dev(ib), dev(xu), dev(xi)).cpu().numpy()
    assert np.array_equal(got, (ps.max(0) + ub[xu]) + ib[xi])
    # sampled pairs: implicit users (S consecutive pairs per user)
    from tensorrec_amd.sparse import PairIndex
    items = rng.integers(0, n_items, (n_users, S)).astype(np.int32)
    ps = rng.standard_normal((Tn, n_users * S)).astype(np.float32)
    idx = PairIndex.make(dev(items.reshape(-1).astype(np.int64)), dev(items.reshape(-1)), S)
    got = T.ops.collapse_tastes(dev(ps), None, dev(ub), dev(ib), idx, idx).cpu().numpy()
    assert np.array_equal(got, (ps.max(0) + np.repeat(ub, S)) + ib[items.reshape(-1)])


@pytest.mark.parametrize("attention", [False, True])
@pytest.mark.parametrize("mode", ["plain", "serial", "sampled", "dense"])
def test_collapse_backward_vs_torch_autograd(T, attention, mode):
    from tensorrec_amd.sparse import PairIndex
    rng = np.random.default_rng(5)
    n_users, n_items, S, Tn = 23, 31, 4, 3
    if mode == "dense":
        shape = (Tn, n_users, n_items)
    elif mode == "sampled":
        shape = (Tn, n_users * S)
    else:
        shape = (Tn, 700)
    p = rng.standard_normal(shape).astype(np.float32)
    p.reshape(Tn, -1)[:2, ::5] = 0.25                               # exact ties between tastes 0 and 1
    a = rng.standard_normal(shape).astype(np.float32) if attention else None
    ub, ib = rng.standard_normal(n_users).astype(np.float32), rng.standard_normal(n_items).astype(np.float32)
    g = rng.standard_normal(shape[1:]).astype(np.float32)
    xu = xi = None
    if mode == "serial":
        xu, xi = rng.integers(0, n_users, 700), rng.integers(0, n_items, 700)
    if mode == "sampled":
        xi = rng.integers(0, n_items, n_users * S)
        xu = np.repeat(np.arange(n_users), S)

    # torch-CPU autograd reference (amax shares tied gradients evenly, like tf.reduce_max)
    tp, tub, tib = (torch.tensor(x, requires_grad=True) for x in (p, ub, ib))
    ta = torch.tensor(a, requires_grad=True) if attention else None
    c = (tp * torch.softmax(ta, dim=0)).sum(0) if attention else torch.amax(tp, dim=0)
    if mode == "dense":
        c = c + tub[:, None] + tib[None, :]
    elif mode != "plain":
        c = c + tub[torch.from_numpy(xu)] + tib[torch.from_numpy(xi)]
    (c * torch.tensor(g)).sum().backward()

    dp, dub, dib = (dev(x).requires_grad_(True) for x in (p, ub, ib))
    da = dev(a).requires_grad_(True) if attention else None
    if mode == "plain":
        out = T.ops.collapse_tastes(dp, da)
    elif mode == "dense":
        out = T.ops.collapse_tastes(dp, da, dub, dib)
    elif mode == "serial":
        out = T.ops.collapse_tastes(dp, da, dub, dib, dev(xu), dev(xi))
    else:
        idx = PairIndex.make(dev(xi.astype(np.int64)), dev(xi.astype(np.int32)), S)
        out = T.ops.collapse_tastes(dp, da, dub, dib, idx, idx)
    assert np.allclose(out.detach().cpu().numpy(), c.detach().numpy(), rtol=1e-5, atol=1e-6)
    (out * dev(g)).sum().backward()
    assert np.allclose(dp.grad.cpu().numpy(), tp.grad.numpy(), rtol=1e-5, atol=1e-6)
    if attention:
        assert np.allclose(da.grad.cpu().numpy(), ta.grad.numpy(), rtol=1e-4, atol=1e-6)
    if mode != "plain":
        assert np.allclose(dub.grad.cpu().numpy(), tub.grad.numpy(), rtol=1e-4, atol=1e-5)
        assert np.allclose(dib.grad.cpu().numpy(), tib.grad.numpy(), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ models
def dummy(T, n_users=60, n_items=90, seed=0):
    inter, uf, itf = T.util.generate_dummy_data(num_users=n_users, num_items=n_items, interaction_density=.08,
                                                num_user_features=40, num_item_features=50, n_features_per_user=6,
                                                n_features_per_item=7, random_state=seed)
    return sp.csr_matrix(inter), sp.csr_matrix(uf), sp.csr_matrix(itf)


def make_pair(T, d, pred, loss, n_tastes, attention, data, tables=None, biased=True):
    from tensorrec_amd.representation_graphs import LinearRepresentationGraph, NormalizedLinearRepresentationGraph
    from tensorrec_amd import prediction_graphs as PG, loss_graphs as LG
    REPR = {"linear": LinearRepresentationGraph, "normalized_linear": NormalizedLinearRepresentationGraph}
    PRED = {"dot": PG.DotProductPredictionGraph, "cosine": PG.CosineSimilarityPredictionGraph,
            "euclidean": PG.EuclideanSimilarityPredictionGraph}
    LOSS = {"rmse": LG.RMSELossGraph, "wmrb": LG.WMRBLossGraph, "balanced_wmrb": LG.BalancedWMRBLossGraph}
    inter, uf, itf = data
    oracle = OracleTensorRec(d, "linear", "linear", pred, loss, biased, n_tastes=n_tastes, attention=attention)
    oracle.init_weights(uf.shape[1], itf.shape[1], np.random.default_rng(42))
    if biased:
        rng = np.random.default_rng(7)
        oracle.weights["user_feature_biases"] = (0.1 * rng.standard_normal((uf.shape[1], 1))).astype(np.float32)
        oracle.weights["item_feature_biases"] = (0.1 * rng.standard_normal((itf.shape[1], 1))).astype(np.float32)
    model = T.TensorRec(n_components=d, n_tastes=n_tastes, prediction_graph=PRED[pred](), loss_graph=LOSS[loss](),
                        attention_graph=REPR[attention]() if attention else None, biased=biased,
                        sampler=T.ReplaySampler(tables) if tables is not None else None, seed=1)
    model.build(uf.shape[1], itf.shape[1])
    assert set(model.get_weights()) == set(oracle.weights)
    model.set_weights(oracle.weights)
    return model, oracle


@pytest.mark.parametrize("pred,loss,n_tastes,attention,d", [
    ("dot", "wmrb", 3, None, 32),
    ("dot", "wmrb", 3, "linear", 32),
    ("dot", "rmse", 2, "linear", 16),
    ("cosine", "balanced_wmrb", 3, "normalized_linear", 20),
    ("euclidean", "wmrb", 2, None, 16),
])
def test_mixture_fit_steps_match_oracle(T, pred, loss, n_tastes, attention, d):
    data = dummy(T)
    inter, uf, itf = data
    S, steps = 11, 3
    tables = None
    if "wmrb" in loss:
        rng = np.random.RandomState(3)
        tables = [O.sample_items(itf.shape[0], uf.shape[0], S, False, rng)[:, 1].reshape(uf.shape[0], S)
                  for _ in range(steps)]
    model, oracle = make_pair(T, d, pred, loss, n_tastes, attention, data, tables)

    p_gpu, p_ref = model.predict(uf, itf), oracle.predict(uf, itf)
    assert np.abs(p_gpu - p_ref).max() <= 1e-4 * np.abs(p_ref).max()          # north_star: 1e-4 relative
    # ranks: equal wherever the oracle's float32 scores are not within rounding distance of a neighbour
    r_gpu = model.predict_rank(uf, itf)
    assert (np.sort(r_gpu, axis=1) == np.arange(1, itf.shape[0] + 1)[None, :]).all()
    assert np.array_equal(r_gpu, O.rank_predictions_exact(p_gpu))

    model._capture = {}
    for t in range(steps):
        model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, alpha=1e-4,
                          n_sampled_items=S if tables else None)
        if t == 0:
            basic, _, grads, pred_serial = oracle.loss_and_grads(inter, uf, itf, 0.0, tables[0] if tables else None)
            cap = model._capture
            assert np.allclose(cap['pred_serial'], pred_serial, rtol=1e-4, atol=1e-4 * np.abs(pred_serial).max())
            assert np.allclose(cap['loss'], basic, rtol=1e-4, atol=1e-5)
            gmax = max(np.abs(g).max() for g in grads.values() if g is not None)
            for k, ref in grads.items():
                if ref is None:
                    continue
                got_g = cap['grads'][k]
                assert got_g is not None and got_g.shape == ref.shape, k
                assert np.abs(got_g - ref).max() <= 1e-4 * gmax, "%s: grad diff %g (gmax %g)" % (
                    k, np.abs(got_g - ref).max(), gmax)
        oracle.step(inter, uf, itf, 0.05, 1e-4, tables[t] if tables else None)
    got = model.get_weights()
    for k, ref in oracle.weights.items():
        assert np.allclose(got[k], ref, rtol=2e-3, atol=5e-3), "%s: max abs diff %g" % (k, np.abs(got[k] - ref).max())


def test_mixture_predict_bit_exact_and_top_k(T):
    """Max over tastes in fp32: per-taste scores are the bit-exact K2 chain, the collapse is an exact max and the bias
    adds keep the reference's order, so the whole matrix equals the C oracle's; predict_top_k (per-taste fused top-k +
    merge) equals the first k ranks."""
    inter, uf, itf = dummy(T, 70, 333, seed=5)
    model, _ = make_pair(T, 24, "dot", "wmrb", 3, None, (inter, uf, itf))
    model.fit(inter, uf, itf, epochs=2, n_sampled_items=7)
    w = model.get_weights()
    v = O.spmm_exact(itf, w["linear_weights_item"])
    ub = O.spmm_exact(uf, w["user_feature_biases"]).reshape(-1)
    ib = O.spmm_exact(itf, w["item_feature_biases"]).reshape(-1)
    us = [O.spmm_exact(uf, w["linear_weights_user_%d" % t]) for t in range(3)]
    ref = np.max(np.stack([O.score_dense_exact(u, v, None, None) for u in us]), axis=0)
    ref = (ref + ub[:, None]) + ib[None, :]
    pred = model.predict(uf, itf)
    assert np.array_equal(pred, ref)
    assert np.array_equal(model.predict_rank(uf, itf), O.rank_predictions_exact(ref))
    for k in (1, 10, 16):
        vals, idx = model.predict_top_k(uf, itf, k=k)
        rv, ri = O.topk_rows(ref, k)
        assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    reprs = model.predict_user_representation(uf)
    assert reprs.shape == (3, 70, 24) and all(np.array_equal(reprs[t], us[t]) for t in range(3))


def test_mixture_top_k_with_duplicate_items_across_tastes(T):
    """Two identical tastes: every item appears in both per-taste lists; the merge must keep one copy."""
    inter, uf, itf = dummy(T, 40, 200, seed=2)
    model, _ = make_pair(T, 16, "dot", "rmse", 2, None, (inter, uf, itf))
    w = model.get_weights()
    w["linear_weights_user_1"] = w["linear_weights_user_0"].copy()
    model.set_weights(w)
    pred = model.predict(uf, itf)
    vals, idx = model.predict_top_k(uf, itf, k=10)
    rv, ri = O.topk_rows(pred, 10)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


@pytest.mark.parametrize("pred", ["dot", "euclidean"])
def test_attention_top_k_matches_oracle_predictions(T, pred):
    """predict_top_k of an attention model (recommendation_graphs.py:98-107 feeding rank order): the k best entries of the
    ORACLE's prediction matrix, in several user batches (user_batch_size below n_users), ties by lower item id."""
    inter, uf, itf = dummy(T, 70, 300, seed=5)
    model, oracle = make_pair(T, 12, pred, "rmse", 3, "linear", (inter, uf, itf))
    ref = oracle.predict(uf, itf)
    for ubs in (None, 16):
        vals, idx = model.predict_top_k(uf, itf, k=7, user_batch_size=ubs)
        rv, ri = O.topk_rows(ref, 7)
        np.testing.assert_allclose(vals, rv, rtol=2e-6, atol=2e-6)         # (expf: libm here, the device's there)
        bad = idx != ri                                                      # ids may only differ inside such a near-tie
        assert np.all(np.abs(np.take_along_axis(ref, idx.astype(np.int64), 1) - rv)[bad] <= 4e-6)
    got = model.predict(uf, itf)
    vals, idx = model.predict_top_k(uf, itf, k=7)
    rv, ri = O.topk_rows(got, 7)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


# ---- API behaviour (test/test_tensorrec.py:299-397 restated) ----------------------------------------------------
@pytest.mark.parametrize("attention", [False, True])
def test_mixture_api_shapes(T, attention):
    from tensorrec_amd.representation_graphs import LinearRepresentationGraph
    inter, uf, itf = T.util.generate_dummy_data(num_users=15, num_items=30, interaction_density=.5, num_user_features=200,
                                                num_item_features=200, n_features_per_user=20, n_features_per_item=20,
                                                pos_int_ratio=.5, random_state=0)
    model = T.TensorRec(n_components=10, n_tastes=3,
                        attention_graph=LinearRepresentationGraph() if attention else None)
    model.fit(inter, uf, itf, epochs=10)
    assert model.predict(uf, itf).shape == (15, 30)
    ranks = model.predict_rank(uf, itf)
    assert ranks.shape == (15, 30) and (ranks > 0).all()
    assert model.predict_user_representation(uf).shape == (3, 15, 10)
    assert model.predict_item_representation(itf).shape == (30, 10)
    assert any(model.predict_user_bias(uf)) and any(model.predict_item_bias(itf))      # test_tensorrec.py:269-275
    sims = model.predict_similar_items(itf, item_ids=[6, 12], n_similar=5)
    assert len(sims) == 2 and all(len(s) == 5 for s in sims)
    if attention:
        assert model.predict_user_attention_representation(uf).shape == (3, 15, 10)
        vals, idx = model.predict_top_k(uf, itf, k=5)          # slab route: softmax-weighted collapse, exact ranks
        rv, ri = O.topk_rows(model.predict(uf, itf), 5)
        assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    else:
        from tensorrec_amd.errors import ModelWithoutAttentionException
        with pytest.raises(ModelWithoutAttentionException):
            model.predict_user_attention_representation(uf)


def test_attention_needs_tastes():
    import tensorrec_amd as T
    from tensorrec_amd.representation_graphs import LinearRepresentationGraph
    with pytest.raises(ValueError):
        T.TensorRec(n_tastes=1, attention_graph=LinearRepresentationGraph())
    with pytest.raises(ValueError):
        T.TensorRec(n_tastes=2, attention_graph=object())
