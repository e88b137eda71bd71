"""User-sharded data-parallel fit: two processes (gloo carrying CUDA tensors; both on the one GPU of the test box) must
reproduce the single-process step on the union of their user shards -- same samples (sampler keyed by global user id),
summed gradients, identical Adam."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _data(identity_users=False):
    rng = np.random.RandomState(0)
    n_u, n_i = 70, 120
    inter = sp.random(n_u, n_i, density=0.08, random_state=rng, format="csr", dtype=np.float32)
    inter.data[:] = 1.0
    uf = sp.hstack([sp.identity(n_u, format="csr", dtype=np.float32),
                    sp.random(n_u, 9, density=0.3, random_state=rng, format="csr", dtype=np.float32)], format="csr")
    if identity_users:               # one-hot user features: under user sharding the user tables' gradient rows are rank-disjoint
        uf = sp.identity(n_u, format="csr", dtype=np.float32)
    itf = sp.identity(n_i, format="csr", dtype=np.float32)
    return inter, uf, itf


def _model(dp, kind="linear"):
    import tensorrec_amd as T
    if kind == "relu_euclid":      # BASELINE.json configs[4] in miniature: ReLU towers + Euclidean similarity + WMRB
        return T.TensorRec(n_components=16, user_repr_graph=T.representation_graphs.ReLURepresentationGraph(),
                           item_repr_graph=T.representation_graphs.ReLURepresentationGraph(),
                           prediction_graph=T.prediction_graphs.EuclideanSimilarityPredictionGraph(),
                           loss_graph=T.loss_graphs.WMRBLossGraph(), seed=5, data_parallel=dp)
    if kind in _SCALAR_LOSSES:
        return T.TensorRec(n_components=16, loss_graph=getattr(T.loss_graphs, kind)(), seed=5, data_parallel=dp)
    return T.TensorRec(n_components=16, loss_graph=T.loss_graphs.BalancedWMRBLossGraph(), seed=5, data_parallel=dp)


_SCALAR_LOSSES = ("RMSELossGraph", "RMSEDenseLossGraph", "SeparationLossGraph", "SeparationDenseLossGraph")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bounds, ret, kind="linear", identity_users=False, min_numel=None, calls=1):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensorrec_amd import sharding
        if min_numel is not None:
            sharding.SHARD_MIN_NUMEL = min_numel
        inter, uf, itf = _data(identity_users)
        if kind in _SCALAR_LOSSES:
            inter = _signed(inter)
        b, e = bounds[rank], bounds[rank + 1]
        model = _model(True, kind)
        for _ in range(calls):
            model.fit_partial(inter[b:e], uf[b:e], itf, epochs=3 // calls, learning_rate=0.05, user_offset=b,
                              n_sampled_items=None if kind in _SCALAR_LOSSES else 20)
        out = model.get_weights()
        out["__plan__"] = dict(model._dp_plan.mode)
        model.dp_sync(optimizer_state=True)
        out["__adam_m_item__"] = model._adam["linear_weights_item"][0].cpu().numpy() if "linear_weights_item" in model._adam else None
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_two_rank_fit_equals_single_process_fit():
    inter, uf, itf = _data()
    single = _model(False)
    single.fit(inter, uf, itf, epochs=3, learning_rate=0.05, n_sampled_items=20)
    ref = single.get_weights()
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, [0, 41, 70], ret), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for k, v in ref.items():
        assert np.array_equal(ret[0][k], ret[1][k]), "ranks diverged on %s" % k       # identical replicas
        if k == "user_feature_biases":
            # Under WMRB a user bias shifts the positive and the sampled scores of that user alike, so its gradient is
            # exactly 0 in exact arithmetic; what Adam sees is rounding noise, which it normalises to +-lr steps (in
            # the reference as well).  Different summation orders give different noise -> only the scale is checked.
            assert np.abs(ret[0][k]).max() <= 3 * 0.05 + 1e-6 and np.abs(v).max() <= 3 * 0.05 + 1e-6
            continue
        # same gradient up to summation order; 0.1 * lr after 3 Adam steps (see test_fit_steps_match_oracle)
        assert np.allclose(ret[0][k], v, rtol=2e-3, atol=5e-3), "%s: %g" % (k, np.abs(ret[0][k] - v).max())
    moved = np.abs(ref["linear_weights_item"] - _initial("linear_weights_item")).max()
    assert moved > 0.05          # the comparison above is not trivially true: weights did move


@pytest.mark.parametrize("calls", [1, 3])
def test_two_rank_fit_rank_disjoint_and_sharded_tables(calls):
    """Identity user features: the user table and the user biases receive gradient only in a rank's own rows -- no exchange, the
    owner steps them (sharding "disjoint"); the item table is reduce-scattered, stepped by row range and all-gathered
    ("sharded": the threshold is lowered for this toy size); the item biases stay all-reduced.  The replicas are identical
    after the end-of-call sync and equal the single-process fit on the union batch; also as three one-epoch calls."""
    inter, uf, itf = _data(identity_users=True)
    single = _model(False)
    single.fit(inter, uf, itf, epochs=3, learning_rate=0.05, n_sampled_items=20)
    ref = single.get_weights()
    ref_m = single._adam["linear_weights_item"][0].cpu().numpy()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), [0, 41, 70], ret, "linear", True, 1000, calls), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    plan = ret[0]["__plan__"]
    assert plan["linear_weights_user_0"] == "disjoint" and plan["user_feature_biases"] == "disjoint"
    assert plan["linear_weights_item"] == "sharded" and plan["item_feature_biases"] == "replicated"
    for k, v in ref.items():
        assert np.array_equal(ret[0][k], ret[1][k]), "ranks diverged on %s" % k
        if k == "user_feature_biases":
            assert np.abs(ret[0][k]).max() <= 3 * 0.05 + 1e-6
            continue
        assert np.allclose(ret[0][k], v, rtol=2e-3, atol=5e-3), "%s: %g" % (k, np.abs(ret[0][k] - v).max())
    # the user table is stepped from ONE rank's gradient: the same sums as the single process makes -> tight agreement
    assert np.allclose(ret[0]["linear_weights_user_0"], ref["linear_weights_user_0"], rtol=1e-5, atol=1e-6)
    # the Adam slots come together on request
    assert np.array_equal(ret[0]["__adam_m_item__"], ret[1]["__adam_m_item__"])
    assert np.allclose(ret[0]["__adam_m_item__"], ref_m, rtol=1e-3, atol=1e-5)


def test_two_rank_fit_relu_euclidean_wmrb():
    """The 8-GPU fit configuration of BASELINE.json (ReLU d=256 + Euclidean + WMRB, users sharded) at toy size."""
    inter, uf, itf = _data()
    single = _model(False, "relu_euclid")
    single.fit(inter, uf, itf, epochs=3, learning_rate=0.05, n_sampled_items=20)
    ref = single.get_weights()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), [0, 33, 70], ret, "relu_euclid"), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for k, v in ref.items():
        assert np.array_equal(ret[0][k], ret[1][k]), "ranks diverged on %s" % k
        if k == "user_feature_biases":
            continue                                   # zero-gradient weight under WMRB (see above)
        assert np.allclose(ret[0][k], v, rtol=2e-3, atol=5e-3), "%s: %g" % (k, np.abs(ret[0][k] - v).max())


def _signed(inter):
    """positive and non-positive interaction values (the separation losses split on the sign, RMSE fits the magnitude)"""
    inter = inter.copy()
    rng = np.random.RandomState(3)
    inter.data[:] = np.where(rng.rand(inter.nnz) < 0.4, -1.0, 1.0) * (0.5 + rng.rand(inter.nnz))
    return inter.astype(np.float32)


@pytest.mark.parametrize("loss", _SCALAR_LOSSES)
def test_two_rank_fit_scalar_losses_equal_single_process_fit(loss):
    """The scalar losses (loss_graphs.py:58-134) under user shards: every rank's loss op all-reduces its sums
    (ops.scalar_loss_group), so both ranks differentiate the ONE scalar of the union batch -- the single-process fit with
    user_batch_size=None -- and the L2 term counts one loss entry, not one per rank."""
    inter, uf, itf = _data()
    inter = _signed(inter)
    single = _model(False, loss)
    single.fit(inter, uf, itf, epochs=3, learning_rate=0.05)
    ref = single.get_weights()
    init = _model(False, loss)
    init.build(uf.shape[1], itf.shape[1])
    init = init.get_weights()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), [0, 41, 70], ret, loss), nprocs=2, join=True)
    assert sorted(ret.keys()) == [0, 1]
    moved = 0.0
    for k, v in ref.items():
        assert np.array_equal(ret[0][k], ret[1][k]), "ranks diverged on %s" % k
        # Adam normalises: a weight whose gradient is rounding noise moves by +-lr whatever the noise is; the weights that carry
        # signal agree to summation order
        close = np.isclose(ret[0][k], v, rtol=2e-3, atol=5e-3)
        assert close.mean() > 0.995, "%s: %.4f of the entries agree, max diff %g" % (k, close.mean(), np.abs(ret[0][k] - v).max())
        moved = max(moved, float(np.abs(v - init[k]).max()))
    assert moved > 0.05


def _initial(name):
    inter, uf, itf = _data()
    m = _model(False)
    m.build(uf.shape[1], itf.shape[1])
    return m.get_weights()[name]


# ---- item-sharded predict_top_k through the public API (BASELINE.json configs[3] in miniature) ---------------------------
def _topk_case(n_items, n_tastes, d=32, graph="cosine"):
    import tensorrec_amd as T
    rng = np.random.RandomState(1)
    n_u = 150
    uf = sp.random(n_u, 30, density=0.2, random_state=rng, format="csr", dtype=np.float32)
    itf = sp.hstack([sp.identity(n_items, format="csr", dtype=np.float32),
                     sp.random(n_items, 5, density=0.3, random_state=rng, format="csr", dtype=np.float32)], format="csr")
    pg = T.prediction_graphs.EuclideanSimilarityPredictionGraph() if graph == "euclidean" else \
        T.prediction_graphs.CosineSimilarityPredictionGraph()
    model = T.TensorRec(n_components=d, n_tastes=n_tastes, seed=11, prediction_graph=pg)
    model.build(uf.shape[1], itf.shape[1])
    w = model.get_weights()
    # (Euclidean: item biases below the distance gaps, so that the per-user certificate holds for most users)
    w["item_feature_biases"] = ((0.002 if graph == "euclidean" else 0.05) *
                                rng.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
    w["user_feature_biases"] = (0.05 * rng.standard_normal(w["user_feature_biases"].shape)).astype(np.float32)
    model.set_weights(w)
    return model, uf, itf


def _topk_worker(rank, world, port, n_items, n_tastes, ret, d=32, graph="cosine", k=10):
    import torch.distributed as dist
    from tensorrec_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if os.environ.get("TREC_TEST_ONE_PASS"):                # the one-pass select + collect, forced on these small tables
            from tensorrec_amd import _native
            _native.set_tuning("filter_scan_one_pass", 2)
            _native.set_tuning("cascade_candidates", 0)         # (the table-driven tail; the default is the candidate lists)
        model, uf, itf = _topk_case(n_items, n_tastes, d, graph)
        b, e = sharding.shard_bounds(n_items, world, rank)
        ret[rank] = model.predict_top_k(uf, itf[b:e], k=k, item_sharded=True, item_offset=b)
        ret["api_route%d" % rank] = dict(model.last_route)
        if graph == "euclidean":
            from tensorrec_amd import ops
            ret["route%d" % rank] = str(ops.LAST_FILTER_STATS.get("route", ""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items,n_tastes,d", [(40000, 1, 32), (3000, 1, 32), (40000, 2, 32), (300000, 1, 64)])
def test_item_sharded_predict_top_k_equals_single_process(n_items, n_tastes, d):
    """Cosine similarity, biased, items sharded over two ranks: both ranks return the single-process result exactly
    (two-stage path with the shared floor for 20,000-item shards, direct fused path for small ones; 300,000 items at
    d = 64: the int8 -> bf16 -> fp32 cascade with both of its floors shared between the shards)."""
    model, uf, itf = _topk_case(n_items, n_tastes, d)
    ref_v, ref_i = model.predict_top_k(uf, itf, k=10)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_topk_worker, args=(2, _free_port(), n_items, n_tastes, ret, d), nprocs=2, join=True)
    for r in (0, 1):
        v, i = ret[r]
        assert np.array_equal(i, ref_i) and np.array_equal(v, ref_v), "rank %d" % r


@pytest.mark.parametrize("n_items,n_tastes", [(60000, 1), (40000, 2)])
def test_item_sharded_euclidean_predict_top_k_takes_the_certified_route(n_items, n_tastes):
    """Euclidean similarity (prediction_graphs.py:84-100) under item shards and with two tastes: every rank runs the dot cascade
    on u.i - r_i / 2 over ITS shard, certifies its own first k and the exact per-shard lists merge -- the single-process result,
    bit for bit (these cases used to leave the fast route)."""
    model, uf, itf = _topk_case(n_items, n_tastes, 32, "euclidean")
    ref_v, ref_i = model.predict_top_k(uf, itf, k=10)
    scores = model.predict(uf, itf)                                   # ... which is the dense prediction's order
    from oracle import oracle as O
    ov, oi = O.topk_rows(scores, 10)
    assert np.array_equal(ref_i, oi) and np.array_equal(ref_v, ov)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_topk_worker, args=(2, _free_port(), n_items, n_tastes, ret, 32, "euclidean"), nprocs=2, join=True)
    for r in (0, 1):
        v, i = ret[r]
        assert np.array_equal(i, ref_i) and np.array_equal(v, ref_v), "rank %d" % r
        assert "euclidean" in ret["route%d" % r], ret["route%d" % r]


def test_item_sharded_cascade_with_the_one_pass_scan(monkeypatch):
    """300,000 items over two ranks with select + collect as ONE scan (forced: these tables are small): the scan's k largest
    entries are what the ranks exchange for the shared floor, the provisional floor comes from the shared tau8."""
    monkeypatch.setenv("TREC_TEST_ONE_PASS", "1")
    model, uf, itf = _topk_case(300000, 1, 64)
    ref_v, ref_i = model.predict_top_k(uf, itf, k=10)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_topk_worker, args=(2, _free_port(), 300000, 1, ret, 64), nprocs=2, join=True)
    for r in (0, 1):
        v, i = ret[r]
        assert np.array_equal(i, ref_i) and np.array_equal(v, ref_v), "rank %d" % r


def test_item_sharded_wide_k_takes_the_cascade_on_every_shard():
    """k = 40 (above the 16 slots of the fused lists) under item shards: every rank runs the wide cascade route on ITS 300,000-item
    shard -- local thresholds, no collective inside the route -- and the exact per-shard lists of 40 merge: the single-process
    result (itself the dense prediction's order, recommendation_graphs.py:73-82 truncated to 40 places), on both ranks."""
    model, uf, itf = _topk_case(600000, 1, 64)
    ref_v, ref_i = model.predict_top_k(uf, itf, k=40)
    assert model.last_route["route"] == "wide_cascade", model.last_route
    scores = model.predict(uf, itf)
    from oracle import oracle as O
    ov, oi = O.topk_rows(scores, 40)
    assert np.array_equal(ref_i, oi) and np.array_equal(ref_v, ov)
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_topk_worker, args=(2, _free_port(), 600000, 1, ret, 64, "cosine", 40), nprocs=2, join=True)
    for r in (0, 1):
        v, i = ret[r]
        assert np.array_equal(i, ref_i) and np.array_equal(v, ref_v), "rank %d" % r
        assert ret["api_route%d" % r]["route"] == "wide_cascade" and ret["api_route%d" % r]["sharded"], ret["api_route%d" % r]
