"""Model-level parity and API tests on the GPU: TensorRec (HIP path) against oracle.model.OracleTensorRec from
IDENTICAL injected weights (the reference exposes no seed), plus the reference's own API-shape tests
(test/test_tensorrec.py, test/test_readme.py) restated for this engine."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import oracle as O
from oracle.model import OracleTensorRec
from parity_util import check_weights_after_adam

pytestmark = pytest.mark.gpu

import tensorrec_amd as T  # noqa: E402
from tensorrec_amd.loss_graphs import (RMSELossGraph, RMSEDenseLossGraph, WMRBLossGraph, BalancedWMRBLossGraph,  # noqa
                                       SeparationLossGraph, SeparationDenseLossGraph, AbstractLossGraph)
from tensorrec_amd.prediction_graphs import (DotProductPredictionGraph, CosineSimilarityPredictionGraph,  # noqa
                                             EuclideanSimilarityPredictionGraph)
from tensorrec_amd.representation_graphs import (LinearRepresentationGraph, NormalizedLinearRepresentationGraph,  # noqa
                                                 ReLURepresentationGraph, FeaturePassThroughRepresentationGraph,
                                                 WeightedFeaturePassThroughRepresentationGraph,
                                                 AbstractRepresentationGraph)

REPR = {"linear": LinearRepresentationGraph, "normalized_linear": NormalizedLinearRepresentationGraph,
        "relu": ReLURepresentationGraph}
PRED = {"dot": DotProductPredictionGraph, "cosine": CosineSimilarityPredictionGraph,
        "euclidean": EuclideanSimilarityPredictionGraph}
LOSS = {"rmse": RMSELossGraph, "wmrb": WMRBLossGraph, "balanced_wmrb": BalancedWMRBLossGraph,
        "rmse_dense": RMSEDenseLossGraph, "separation": SeparationLossGraph, "separation_dense": SeparationDenseLossGraph}


def dummy(n_users=60, n_items=90, seed=0):
    inter, uf, itf = T.util.generate_dummy_data(num_users=n_users, num_items=n_items, interaction_density=.08,
                                                num_user_features=40, num_item_features=50, n_features_per_user=6,
                                                n_features_per_item=7, random_state=seed)
    return sp.csr_matrix(inter), sp.csr_matrix(uf), sp.csr_matrix(itf)


def make_pair(d, user_repr, item_repr, pred, loss, biased, data, sample_tables=None):
    inter, uf, itf = data
    oracle = OracleTensorRec(d, user_repr, item_repr, pred, loss, biased)
    oracle.init_weights(uf.shape[1], itf.shape[1], np.random.default_rng(42))
    if biased:      # non-zero biases so the bias path is actually exercised from step 1
        rng = np.random.default_rng(7)
        oracle.weights["user_feature_biases"] = (0.1 * rng.standard_normal((uf.shape[1], 1))).astype(np.float32)
        oracle.weights["item_feature_biases"] = (0.1 * rng.standard_normal((itf.shape[1], 1))).astype(np.float32)
    sampler = T.ReplaySampler(sample_tables) if sample_tables is not None else None
    model = T.TensorRec(n_components=d, user_repr_graph=REPR[user_repr](), item_repr_graph=REPR[item_repr](),
                        prediction_graph=PRED[pred](), loss_graph=LOSS[loss](), biased=biased, sampler=sampler, seed=1)
    model.build(uf.shape[1], itf.shape[1])
    model.set_weights(_rename(oracle.weights))
    return model, oracle


def _rename(w):
    out = {}
    for k, v in w.items():
        if k.endswith("_user"):
            k = k + "_0"                # node_name_ending 'user_0' (tensorrec.py:344)
        out[k] = v
    return out


@pytest.mark.parametrize("user_repr,item_repr,pred,loss,biased,d", [
    ("linear", "linear", "dot", "rmse", True, 100),                 # the reference's defaults (BASELINE config 1)
    ("linear", "linear", "dot", "rmse", False, 32),
    ("linear", "linear", "dot", "wmrb", True, 64),                  # BASELINE config 2 shape
    ("normalized_linear", "linear", "cosine", "balanced_wmrb", True, 20),
    ("relu", "relu", "euclidean", "wmrb", True, 16),                # BASELINE config 5 shape
    ("linear", "relu", "euclidean", "rmse", False, 12),
    ("linear", "linear", "dot", "rmse_dense", True, 16),            # dense + separation losses
    ("linear", "linear", "dot", "separation", True, 16),
    ("normalized_linear", "linear", "cosine", "separation_dense", False, 12),
])
def test_fit_steps_match_oracle(user_repr, item_repr, pred, loss, biased, d):
    data = dummy()
    inter, uf, itf = data
    S, steps = 11, 3
    tables = None
    if "wmrb" in loss:
        rng = np.random.RandomState(3)
        tables = [O.sample_items(itf.shape[0], uf.shape[0], S, False, rng)[:, 1].reshape(uf.shape[0], S)
                  for _ in range(steps)]
    model, oracle = make_pair(d, user_repr, item_repr, pred, loss, biased, data, tables)

    # forward parity before any training
    p_gpu = model.predict(uf, itf)
    p_ref = oracle.predict(uf, itf)
    scale = np.abs(p_ref).max()
    assert np.abs(p_gpu - p_ref).max() <= 1e-4 * scale          # north_star: 1e-4 relative on float scores

    model._capture = {}
    for t in range(steps):
        model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, alpha=1e-4,
                          n_sampled_items=S if tables else None)
        if t == 0:
            # first step, identical weights on both sides: loss vector, serial predictions and RAW gradients
            # (before Adam, without the L2 term) must agree.  Tolerance: 1e-4 of the largest gradient entry of the
            # model -- some gradients are exactly zero in exact arithmetic (e.g. user biases under WMRB cancel
            # between positives and samples), so a per-tensor relative bound would compare noise with noise.
            basic, _, grads, pred_serial = oracle.loss_and_grads(inter, uf, itf, 0.0, tables[0] if tables else None)
            cap = model._capture
            assert np.allclose(cap['pred_serial'], pred_serial, rtol=1e-4, atol=1e-4 * np.abs(pred_serial).max())
            assert np.allclose(cap['loss'], basic, rtol=1e-4, atol=1e-5)
            gmax = max(np.abs(g).max() for g in grads.values() if g is not None)
            for k, ref in _rename(grads).items():
                if ref is None:
                    continue
                got_g = cap['grads'][k]
                assert got_g is not None and got_g.shape == ref.shape, k
                assert np.abs(got_g - ref).max() <= 1e-4 * gmax, "%s: grad diff %g (gmax %g)" % (
                    k, np.abs(got_g - ref).max(), gmax)
            grads0_gpu, grads0_ref = dict(cap['grads']), grads
        oracle.step(inter, uf, itf, 0.05, 1e-4, tables[t] if tables else None)
    got = model.get_weights()
    # tests/parity_util.py: 1e-4 * lr per step wherever the first-step gradient stands clear of the fp32 summation
    # noise, proportionally looser below (Adam steps by ~lr * g / |g|).  Exempt -- gradient provably ZERO in exact
    # arithmetic: user_feature_biases under WMRB / BalancedWMRB (b_u cancels inside every hinge 1 - y_ui + y_us).
    exempt = ("user_feature_biases",) if "wmrb" in loss else ()
    rep = check_weights_after_adam(got, _rename(oracle.weights), grads0_gpu, _rename(grads0_ref), 0.05, steps,
                                   exempt=exempt, label="%s/%s/%s/%s" % (user_repr, item_repr, pred, loss))
    print("weights after %d steps (max |dw|, share beyond 1e-4 lr/step): %s" % (steps, rep))
    p_gpu = model.predict(uf, itf)
    p_ref = oracle.predict(uf, itf)
    assert np.abs(p_gpu - p_ref).max() <= 2e-2 * max(1.0, np.abs(p_ref).max())


def test_predict_and_rank_bit_exact_from_same_weights():
    """Linear + DotProduct + biases in fp32: K1 (fmaf in CSR order) -> K2 (fmaf chain over k) -> bias adds in the
    reference's order.  From the same weights the whole score matrix is bit-identical to the C oracle, hence so are
    the ranks (north_star: 'bit-exact for predicted ranks')."""
    inter, uf, itf = dummy(70, 333, seed=5)
    model, oracle = make_pair(100, "linear", "linear", "dot", "rmse", True, (inter, uf, itf))
    model.fit(inter, uf, itf, epochs=2)                       # move the weights off their initial values
    w = model.get_weights()
    u = O.spmm_exact(uf, w["linear_weights_user_0"])
    v = O.spmm_exact(itf, w["linear_weights_item"])
    ub = O.spmm_exact(uf, w["user_feature_biases"]).reshape(-1)
    ib = O.spmm_exact(itf, w["item_feature_biases"]).reshape(-1)
    ref = O.score_dense_exact(u, v, ub, ib)
    pred = model.predict(uf, itf)
    assert pred.dtype == np.float32 and np.array_equal(pred, ref)
    ranks = model.predict_rank(uf, itf)
    assert ranks.dtype == np.int32 and np.array_equal(ranks, O.rank_predictions_exact(ref))
    vals, idx = model.predict_top_k(uf, itf, k=10)
    rv, ri = O.topk_rows(ref, 10)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
    assert np.array_equal(model.predict_user_representation(uf), u)
    assert np.array_equal(model.predict_item_representation(itf), v)
    assert np.array_equal(model.predict_user_bias(uf), ub)
    assert np.array_equal(model.predict_item_bias(itf), ib)


def test_bf16_precision_mode_reports_rank_agreement():
    inter, uf, itf = dummy(64, 500, seed=6)
    m32 = T.TensorRec(n_components=64, seed=3)
    m32.fit(inter, uf, itf, epochs=2)
    m16 = T.TensorRec(n_components=64, precision='bf16', seed=3)
    m16.build(uf.shape[1], itf.shape[1])
    m16.set_weights(m32.get_weights())
    p32, p16 = m32.predict(uf, itf), m16.predict(uf, itf)
    scale = np.abs(p32).max()
    assert np.abs(p32 - p16).max() <= 2e-2 * scale            # bf16 operands: ~2^-8 per element
    _, i32 = m32.predict_top_k(uf, itf, k=10)
    _, i16 = m16.predict_top_k(uf, itf, k=10)
    overlap = np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(i32, i16)])
    assert overlap > 0.8


# ---- API behaviour (test/test_tensorrec.py, test/test_readme.py restated) -------------------------------------
def test_readme_flow_and_shapes():
    inter, uf, itf = T.util.generate_dummy_data(num_users=100, num_items=150, interaction_density=.05, random_state=0)
    model = T.TensorRec()
    model.fit(inter, uf, itf, epochs=5, verbose=True)
    predictions = model.predict(user_features=uf, item_features=itf)
    ranks = model.predict_rank(user_features=uf, item_features=itf)
    assert predictions.shape == (100, 150) and ranks.shape == (100, 150)
    assert (ranks > 0).all() and (np.sort(ranks, axis=1) == np.arange(1, 151)[None, :]).all()
    assert model.predict_user_representation(uf).shape == (100, 100)
    assert model.predict_item_representation(itf).shape == (150, 100)
    assert model.predict_user_bias(uf).shape == (100,) and model.predict_item_bias(itf).shape == (150,)
    assert np.abs(model.predict_user_bias(uf)).sum() > 0          # biases are trained (test_tensorrec.py:269-275)
    sims = model.predict_similar_items(itf, item_ids=[6, 12], n_similar=5)
    assert len(sims) == 2 and len(sims[0]) == 5
    assert all(0 <= i < 150 for i, _ in sims[0]) and [s for _, s in sims[0]] == sorted((s for _, s in sims[0]), reverse=True)
    cos = T.TensorRec(n_components=10, prediction_graph=T.prediction_graphs.CosineSimilarityPredictionGraph(), seed=0)
    cos.fit(inter, uf, itf, epochs=1)
    assert cos.predict_similar_items(itf, item_ids=[6], n_similar=3)[0][0][0] == 6     # cosine: an item is its own nearest


def test_unfit_and_unbiased_errors():
    inter, uf, itf = dummy()
    model = T.TensorRec(n_components=10, biased=False)
    for name, args in (("predict", (uf, itf)), ("predict_rank", (uf, itf)), ("predict_user_representation", (uf,)),
                       ("predict_item_representation", (itf,)), ("predict_user_bias", (uf,)),
                       ("predict_similar_items", (itf, [1], 2))):
        with pytest.raises(T.errors.ModelNotFitException):
            getattr(model, name)(*args)
    model.fit(inter, uf, itf, epochs=1)
    with pytest.raises(T.errors.ModelNotBiasedException):
        model.predict_user_bias(uf)
    with pytest.raises(T.errors.ModelNotBiasedException):
        model.predict_item_bias(itf)
    with pytest.raises(T.errors.ModelWithoutAttentionException):
        model.predict_user_attention_representation(uf)


def test_fit_argument_checks_and_batching():
    inter, uf, itf = dummy()
    model = T.TensorRec(n_components=8, loss_graph=WMRBLossGraph())
    with pytest.raises(ValueError):
        model.fit(inter, uf, itf, epochs=1)                               # sample-based loss needs n_sampled_items
    with pytest.raises(ValueError):
        model.fit(inter, uf, itf, epochs=1, n_sampled_items=0)
    model.fit(inter, uf, itf, epochs=2, n_sampled_items=10, user_batch_size=25)      # 3 user batches
    assert model.predict(uf, itf).shape == (60, 90)
    with pytest.raises(T.errors.BatchNonSparseInputException):
        T.TensorRec(n_components=8).fit(inter.toarray(), uf, itf, epochs=1, user_batch_size=10)
    with pytest.raises(ValueError):
        T.TensorRec(n_components=8).fit([inter, inter], [uf], itf, epochs=1)          # batch counts differ
    with pytest.raises(ValueError):
        T.TensorRec(n_components=8).fit(inter.toarray(), uf, itf, epochs=1)           # not sparse
    # lists of pre-batched inputs
    m2 = T.TensorRec(n_components=8)
    m2.fit([inter[:30], inter[30:]], [uf[:30], uf[30:]], itf, epochs=1)
    assert m2.predict(uf, itf).shape == (60, 90)


@pytest.mark.parametrize("loss,kw", [(RMSELossGraph, {}), (RMSEDenseLossGraph, {}), (WMRBLossGraph, {"n_sampled_items": 10}),
                                     (BalancedWMRBLossGraph, {"n_sampled_items": 10}), (SeparationLossGraph, {}),
                                     (SeparationDenseLossGraph, {})])
@pytest.mark.parametrize("biased", [True, False])
def test_loss_graphs_run(loss, kw, biased):
    """test/test_loss_graphs.py:17-47 (smoke) -- plus: the loss must actually go down."""
    inter, uf, itf = T.util.generate_dummy_data_with_indicator(num_users=10, num_items=12, interaction_density=.5,
                                                               seed=0)
    model = T.TensorRec(n_components=10, loss_graph=loss(), biased=biased, seed=0)
    model.fit(inter, uf, itf, epochs=5, **kw)
    assert np.isfinite(model.predict(uf, itf)).all()


@pytest.mark.parametrize("u,i,d", [(LinearRepresentationGraph, LinearRepresentationGraph, 50),
                                   (NormalizedLinearRepresentationGraph, LinearRepresentationGraph, 50),
                                   (LinearRepresentationGraph, NormalizedLinearRepresentationGraph, 50),
                                   (ReLURepresentationGraph, ReLURepresentationGraph, 20),
                                   (FeaturePassThroughRepresentationGraph, NormalizedLinearRepresentationGraph, 36),
                                   (WeightedFeaturePassThroughRepresentationGraph, NormalizedLinearRepresentationGraph, 36),
                                   (LinearRepresentationGraph, FeaturePassThroughRepresentationGraph, 48)])
def test_representation_graphs_run(u, i, d):
    """test/test_representation_graphs.py:14-35: 15 x 30 indicator data, n_components sized for the pass-throughs."""
    inter, uf, itf = T.util.generate_dummy_data_with_indicator(num_users=15 * 2, num_items=20 * 2,
                                                               interaction_density=.5, seed=1)
    # user features: 36 columns, item features: 48 columns
    model = T.TensorRec(n_components=d, user_repr_graph=u(), item_repr_graph=i(), seed=0)
    model.fit(inter, uf, itf, epochs=10)
    assert model.predict(uf, itf).shape == (30, 40)


def test_passthrough_dimension_mismatch_raises():
    inter, uf, itf = dummy()
    model = T.TensorRec(n_components=5, user_repr_graph=FeaturePassThroughRepresentationGraph())
    with pytest.raises(ValueError):
        model.fit(inter, uf, itf, epochs=1)


def test_custom_graphs_plug_in():
    """test/test_readme.py:36-108: user-defined representation (tanh) and loss (MAE) subclasses."""
    from tensorrec_amd import ops
    from tensorrec_amd.framework import Variable, random_normal

    class TanhRepresentationGraph(AbstractRepresentationGraph):
        def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
            w = Variable(lambda: random_normal([n_features, n_components], stddev=.5),
                         name='tanh_weights_%s' % node_name_ending)
            return torch.tanh(ops.sparse_dense_matmul(tf_features, w)), [w]

    class SimpleLossGraph(AbstractLossGraph):
        def connect_loss_graph(self, tf_prediction_serial, tf_interactions_serial, **kwargs):
            return torch.mean(torch.abs(tf_interactions_serial - tf_prediction_serial))

    inter, uf, itf = dummy()
    model = T.TensorRec(n_components=12, user_repr_graph=TanhRepresentationGraph(),
                        item_repr_graph=TanhRepresentationGraph(), loss_graph=SimpleLossGraph(), seed=0)
    model.fit(inter, uf, itf, epochs=1)
    before = np.abs(inter.toarray()[inter.toarray() != 0] - model.predict(uf, itf)[inter.toarray() != 0]).mean()
    model.fit(inter, uf, itf, epochs=30, learning_rate=0.01)
    after = np.abs(inter.toarray()[inter.toarray() != 0] - model.predict(uf, itf)[inter.toarray() != 0]).mean()
    assert after < before
    assert set(model.get_weights()) >= {"tanh_weights_user_0", "tanh_weights_item"}


def test_training_reduces_wmrb_loss_and_ranks_positives_higher():
    """End-to-end sanity with the DEVICE sampler: WMRB training must push positives up the ranking."""
    rng = np.random.RandomState(0)
    n_u, n_i = 200, 300
    inter = sp.random(n_u, n_i, density=0.03, random_state=rng, format="csr", dtype=np.float32)
    inter.data[:] = 1.0
    uf, itf = sp.identity(n_u, format="csr", dtype=np.float32), sp.identity(n_i, format="csr", dtype=np.float32)
    model = T.TensorRec(n_components=16, loss_graph=WMRBLossGraph(), seed=0)
    model.fit(inter, uf, itf, epochs=1, n_sampled_items=30)
    r0 = model.predict_rank(uf, itf)[inter.nonzero()].mean()
    model.fit(inter, uf, itf, epochs=40, n_sampled_items=30, learning_rate=0.05)
    r1 = model.predict_rank(uf, itf)[inter.nonzero()].mean()
    assert r1 < 0.5 * r0, (r0, r1)


# ---- persistence (test/test_tensorrec.py:399-458 restated) ------------------------------------------------------
@pytest.mark.parametrize("n_tastes", [1, 3])
def test_save_and_load_model(tmp_path, n_tastes):
    inter, uf, itf = T.util.generate_dummy_data(num_users=15, num_items=30, interaction_density=.5, num_user_features=200,
                                                num_item_features=200, n_features_per_user=20, n_features_per_item=20,
                                                pos_int_ratio=.5, random_state=0)
    with pytest.raises(T.errors.ModelNotFitException):
        T.TensorRec(n_components=10).save_model(str(tmp_path / "unfit"))
    model = T.TensorRec(n_components=10, n_tastes=n_tastes, seed=4)
    model.fit(inter, uf, itf, epochs=10)
    predictions, ranks = model.predict(uf, itf), model.predict_rank(uf, itf)
    directory = str(tmp_path / "model" / "nested")
    model.save_model(directory_path=directory)
    assert (predictions == model.predict(uf, itf)).all() and (ranks == model.predict_rank(uf, itf)).all()
    new_model = T.TensorRec.load_model(directory_path=directory)
    assert (predictions == new_model.predict(uf, itf)).all() and (ranks == new_model.predict_rank(uf, itf)).all()
    # the optimiser state travels too: one more epoch on both gives identical weights (the RMSE step is deterministic)
    assert (new_model._opt_step, new_model._sample_step) == (model._opt_step, model._sample_step)
    model.fit_partial(inter, uf, itf, epochs=1)
    new_model.fit_partial(inter, uf, itf, epochs=1)
    a, b = model.get_weights(), new_model.get_weights()
    assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)


def test_fit_and_eval_and_metrics_from_gpu_ranks():
    from oracle import eval_dense
    inter, uf, itf = T.util.generate_dummy_data_with_indicator(num_users=10, num_items=12, interaction_density=.5,
                                                               seed=1)
    model = T.TensorRec(n_components=10, seed=0)
    out = T.eval.fit_and_eval(model, uf, itf, inter, inter, {"epochs": 10}, recall_k=5, precision_k=5, ndcg_k=5)
    assert len(out) == 6 and out[:3] == out[3:] and all(0.0 <= v <= 1.0 for v in out)
    ranks = model.predict_rank(uf, itf)
    for name in ("precision_at_k", "recall_at_k", "ndcg_at_k"):
        np.testing.assert_allclose(getattr(T.eval, name)(ranks, inter, k=5), getattr(eval_dense, name)(ranks, inter, k=5),
                                   rtol=1e-12)
    ndcg5, ndcg10, ndcg20 = (np.mean(T.eval.ndcg_at_k(ranks, inter, k=k)) for k in (5, 10, 20))
    assert ndcg10 >= ndcg5 and ndcg20 == ndcg10 and ndcg20 < 1                       # test/test_eval.py:80-83
    assert T.eval.f1_score_at_k(ranks, inter, k=5) is not None


@pytest.mark.parametrize("n_tastes", [1, 2])
def test_rank_of_interactions_equals_rank_matrix_entries(n_tastes):
    """Device-side ranks of the positive pairs == predict_rank()[rows, cols], tile boundaries included; the metrics
    computed from them equal the metrics from the full matrix."""
    inter, uf, itf = dummy(150, 333, seed=9)
    model = T.TensorRec(n_components=16, n_tastes=n_tastes, seed=2)
    model.fit(inter, uf, itf, epochs=3)
    ranks = model.predict_rank(uf, itf)
    for batch in (None, 64, 7):
        pr = model.predict_rank_of_interactions(uf, itf, inter, user_batch_size=batch)
        coo = sp.csr_matrix(inter).tocoo()
        pos = coo.data > 0
        assert np.array_equal(pr.rows, coo.row[pos]) and np.array_equal(pr.ranks, ranks[coo.row[pos], coo.col[pos]])
    for f in (T.eval.precision_at_k, T.eval.recall_at_k, T.eval.ndcg_at_k):
        for preserve in (False, True):
            np.testing.assert_array_equal(f(pr, inter, k=10, preserve_rows=preserve),
                                          f(ranks, inter, k=10, preserve_rows=preserve))


def test_fit_from_tfrecords_and_datasets(tmp_path):
    """test/test_tensorrec.py:169-173 and :363-397: TFRecord paths and standard-format datasets are accepted wherever
    scipy matrices are, and give the same model."""
    from tensorrec_amd.input_utils import write_tfrecord_from_sparse_matrix, create_tensorrec_dataset_from_sparse_matrix
    inter, uf, itf = T.util.generate_dummy_data(num_users=15, num_items=30, interaction_density=.5, num_user_features=200,
                                                num_item_features=200, n_features_per_user=20, n_features_per_item=20,
                                                pos_int_ratio=.5, random_state=0)
    paths = [write_tfrecord_from_sparse_matrix(str(tmp_path / name), m)
             for name, m in (("interactions.tfrecord", inter), ("user_features.tfrecord", uf),
                             ("item_features.tfrecord", itf))]
    a = T.TensorRec(n_components=10, seed=1)
    a.fit(inter, uf, itf, epochs=5)
    b = T.TensorRec(n_components=10, seed=1)
    b.fit(paths[0], paths[1], paths[2], epochs=5)
    c = T.TensorRec(n_components=10, seed=1)
    c.fit(*[create_tensorrec_dataset_from_sparse_matrix(m) for m in (inter, uf, itf)], epochs=5)
    pa = a.predict(uf, itf)
    assert np.array_equal(pa, b.predict(paths[1], paths[2])) and np.array_equal(pa, c.predict(uf, itf))
    assert np.array_equal(a.predict_rank(uf, itf), b.predict_rank(paths[1], paths[2]))


@pytest.mark.parametrize("pred,loss,biased,d,S", [("dot", "wmrb", True, 128, 100), ("cosine", "balanced_wmrb", True, 64, 37),
                                                  ("dot", "wmrb", False, 100, 200), ("dot", "balanced_wmrb", True, 256, 8)])
def test_fused_wmrb_step_equals_unfused(pred, loss, biased, d, S):
    """The one-pass WMRB step (csrc/wmrb_fused.hip) against the composed path (serial scores -> loss kernels -> autograd):
    serial predictions, loss vector and gradients equal up to summation order."""
    inter, uf, itf = dummy(150, 333, seed=4)
    inter = sp.csr_matrix(inter)
    inter[7, :] = 0                                   # a user without interactions
    inter.eliminate_zeros()
    rng = np.random.RandomState(5)
    tables = [O.sample_items(itf.shape[0], uf.shape[0], S, False, rng)[:, 1].reshape(uf.shape[0], S)]
    caps = []
    for fused in (1, 0):
        T._native.set_tuning("wmrb_fused", fused)
        try:
            model = T.TensorRec(n_components=d, prediction_graph=PRED[pred](), loss_graph=LOSS[loss](), biased=biased,
                                sampler=T.ReplaySampler(tables), seed=3)
            model.build(uf.shape[1], itf.shape[1])
            if biased:
                w = model.get_weights()
                r2 = np.random.default_rng(7)
                w["user_feature_biases"] = (0.1 * r2.standard_normal(w["user_feature_biases"].shape)).astype(np.float32)
                w["item_feature_biases"] = (0.1 * r2.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
                model.set_weights(w)
            model._capture = {}
            model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, alpha=1e-4, n_sampled_items=S)
            caps.append((model._capture, model.get_weights()))
        finally:
            T._native.set_tuning("wmrb_fused", 1)
    (a, wa), (b, wb) = caps
    assert np.allclose(a['loss'], b['loss'], rtol=1e-5, atol=1e-6)       # hinge sums: sequential vs lane-split order
    assert np.allclose(a['pred_serial'], b['pred_serial'], rtol=1e-5, atol=1e-6)    # DPP rotations vs xor butterfly
    gmax = max(np.abs(g).max() for g in b['grads'].values() if g is not None)
    for k, gb in b['grads'].items():
        if gb is None:
            assert a['grads'][k] is None or not np.abs(a['grads'][k]).any(), k
            continue
        assert np.abs(a['grads'][k] - gb).max() <= 2e-5 * gmax, "%s: %g (gmax %g)" % (k, np.abs(a['grads'][k] - gb).max(), gmax)
    for k in wa:
        if k == "user_feature_biases":
            continue                                  # zero-gradient weight under WMRB: Adam amplifies rounding noise
        assert np.allclose(wa[k], wb[k], rtol=1e-3, atol=2e-3), k


@pytest.mark.parametrize("learned", [False, True])
def test_fit_step_on_the_binned_drop_zero_route_equals_the_oracle(learned):
    """The route the 1M x 1M fit takes -- >= 2^22 sampled pairs: grouped by item through the rank-free binned partition, pairs whose
    coefficient is 0 left out (ops.wmrb_fused_step) -- against oracle/model.py (loss_graphs.py:153-180, tensorrec.py:487-489): one
    optimiser step through the public API on 45,000 users x 20,000 items x S = 100; serial predictions, loss vector, raw
    gradients at 1e-4 of the largest, weights under the Adam-aware bar.  learned=True: a state in which most samples violate no
    margin, so most pairs really are dropped."""
    import bench_records as BR
    from tensorrec_amd import ops, _native as N
    assert N.load().trec_get_tuning(b"group_pairs_binned", 1) and N.load().trec_get_tuning(b"group_pairs_drop_zero", 1)
    ops.KERNEL_EVENTS = []                                 # (statistics on: how many pairs the item side kept)
    try:
        rec, _ = BR.parity_fit_record(20_000, 32, n_users=45_000, per_user=5, n_sampled=100, learned=learned, expect_route="binned")
        kept = int(ops.LAST_FUSED_STATS["sampled_pairs_kept"].item())
    finally:
        ops.KERNEL_EVENTS = None
    assert rec["route"] == {"grouping": "binned", "drop_zero": True, "sampled_pairs": 4_500_000}, rec["route"]
    import json
    assert rec["green"], json.dumps({k_: v for k_, v in rec.items() if k_ != "workload"})
    if learned:
        assert 0 < kept < 0.5 * 4_500_000, kept           # most coefficients are zero and were dropped
    else:
        assert kept > 0.9 * 4_500_000, kept


@pytest.mark.parametrize("pred,loss,biased,d,S,dense_g", [
    ("euclidean", "wmrb", True, 256, 300, 1), ("euclidean", "balanced_wmrb", True, 64, 37, 0), ("dot", "wmrb", True, 128, 320, 1),
    ("cosine", "wmrb", False, 100, 300, 0), ("euclidean", "wmrb", False, 36, 50, 1), ("dot", "balanced_wmrb", True, 512, 280, 0)])
def test_tiled_wmrb_step_equals_unfused(pred, loss, biased, d, S, dense_g):
    """The tiled one-pass WMRB step (csrc/wmrb_tiled.hip: item rows streamed twice, scores in LDS; dot / cosine / Euclidean; item
    side through the dense G^T . U GEMM or through the grouped gathers) against the composed path (serial scores -> loss kernels ->
    autograd, prediction_graphs.py:52-55 / :70-72 / :105-117, loss_graphs.py:153-227): serial predictions, loss vector, raw
    gradients and weights equal up to summation order."""
    from tensorrec_amd import ops
    inter, uf, itf = dummy(150, 333, seed=4)
    inter = sp.csr_matrix(inter)
    inter[7, :] = 0                                   # a user without interactions
    inter.eliminate_zeros()
    rng = np.random.RandomState(5)
    tables = [O.sample_items(itf.shape[0], uf.shape[0], S, False, rng)[:, 1].reshape(uf.shape[0], S)]
    caps = []
    for tiled in (1, 0):
        T._native.set_tuning("wmrb_fused", 0)
        T._native.set_tuning("wmrb_tiled", tiled)
        T._native.set_tuning("wmrb_dense_g", dense_g)
        ops.LAST_FUSED_STATS.pop("route", None)
        try:
            model = T.TensorRec(n_components=d, prediction_graph=PRED[pred](), loss_graph=LOSS[loss](), biased=biased,
                                sampler=T.ReplaySampler(tables), seed=3)
            model.build(uf.shape[1], itf.shape[1])
            if biased:
                w = model.get_weights()
                r2 = np.random.default_rng(7)
                w["user_feature_biases"] = (0.1 * r2.standard_normal(w["user_feature_biases"].shape)).astype(np.float32)
                w["item_feature_biases"] = (0.1 * r2.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
                model.set_weights(w)
            model._capture = {}
            model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, alpha=1e-4, n_sampled_items=S)
            caps.append((model._capture, model.get_weights(), ops.LAST_FUSED_STATS.get("route")))
        finally:
            T._native.set_tuning("wmrb_fused", 1)
            T._native.set_tuning("wmrb_tiled", 1)
            T._native.set_tuning("wmrb_dense_g", 1)
    (a, wa, route_a), (b, wb, route_b) = caps
    assert route_a == ("tiled+dense_g" if dense_g else "tiled+grouped") and route_b is None, (route_a, route_b)
    assert np.allclose(a['loss'], b['loss'], rtol=1e-5, atol=1e-6)
    assert np.allclose(a['pred_serial'], b['pred_serial'], rtol=1e-5, atol=1e-6)
    gmax = max(np.abs(g).max() for g in b['grads'].values() if g is not None)
    for k, gb in b['grads'].items():
        if gb is None:
            assert a['grads'][k] is None or not np.abs(a['grads'][k]).any(), k
            continue
        assert np.abs(a['grads'][k] - gb).max() <= 2e-5 * gmax, "%s: %g (gmax %g)" % (k, np.abs(a['grads'][k] - gb).max(), gmax)
    for k in wa:
        if k == "user_feature_biases":
            continue                                  # zero-gradient weight under WMRB: Adam amplifies rounding noise
        assert np.allclose(wa[k], wb[k], rtol=1e-3, atol=2e-3), k


def test_tiled_wmrb_step_vs_oracle_with_long_rows():
    """The tiled step against oracle/model.py directly: Euclidean scores, S = 1,200 of 3,000 items, users with up to 700
    interactions (rows far beyond the register kernel), negative interactions among them (no loss term, loss_graphs.py:160-163),
    ReLU representations; one optimiser step, the bar of the config records (1e-4)."""
    import bench_records as BR
    from tensorrec_amd import ops
    rng = np.random.default_rng(3)
    n_users, n_items, d, S = 64, 3000, 64, 1200
    rows, cols, vals = [], [], []
    for u_ in range(n_users):
        n = int(rng.integers(1, 700)) if u_ % 3 else 0
        c = rng.choice(n_items, size=n, replace=False)
        rows += [u_] * n
        cols += list(c)
        vals += list(np.where(rng.random(n) < 0.8, 1.0, -1.0))
    inter = sp.csr_matrix((np.array(vals, np.float32), (rows, cols)), shape=(n_users, n_items))
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = BR._side_features(n_items, 40, 3, rng)
    table = np.stack([rng.permutation(n_items)[:S] for _ in range(n_users)]).astype(np.int64)
    oracle = OracleTensorRec(d, "relu", "relu", "euclidean", "wmrb", True)
    oracle.init_weights(uf.shape[1], itf.shape[1], np.random.default_rng(42))

    def mk(tables):
        return T.TensorRec(n_components=d, user_repr_graph=ReLURepresentationGraph(), item_repr_graph=ReLURepresentationGraph(),
                           prediction_graph=EuclideanSimilarityPredictionGraph(), loss_graph=WMRBLossGraph(), seed=0,
                           sampler=T.ReplaySampler(tables))
    ops.LAST_FUSED_STATS.pop("route", None)
    rec = BR._one_step_parity(mk, oracle, inter, uf, itf, table, 0.01, 1e-5, S, 1e-4)
    assert ops.LAST_FUSED_STATS.get("route") == "tiled+dense_g"
    assert rec["green"], rec


@pytest.mark.parametrize("pred,loss,biased,d", [("dot", "rmse_dense", True, 16), ("dot", "separation_dense", True, 24),
                                                 ("cosine", "rmse_dense", False, 12), ("cosine", "separation_dense", True, 130)])
def test_factored_dense_loss_equals_materialised(pred, loss, biased, d):
    """RMSEDense / SeparationDense (loss_graphs.py:62-72, :100-134) from the Gram matrices of the factors (csrc/loss_dense.hip,
    "factored": no [n_users, n_items] tensor) against the same losses as streaming reductions over the materialised prediction:
    loss value, raw gradients and weights after the step.  (Both forms are held to the oracle in test_fit_steps_match_oracle.)"""
    inter, uf, itf = dummy(150, 333, seed=4)
    caps = []
    for factored in (1, 0):
        T._native.set_tuning("dense_loss_factored", factored)
        try:
            model = T.TensorRec(n_components=d, prediction_graph=PRED[pred](), loss_graph=LOSS[loss](), biased=biased, seed=3)
            model.build(uf.shape[1], itf.shape[1])
            if biased:
                w = model.get_weights()
                r2 = np.random.default_rng(7)
                w["user_feature_biases"] = (0.1 * r2.standard_normal(w["user_feature_biases"].shape)).astype(np.float32)
                w["item_feature_biases"] = (0.1 * r2.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
                model.set_weights(w)
            model._capture = {}
            model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, alpha=1e-4)
            caps.append((model._capture, model.get_weights()))
        finally:
            T._native.set_tuning("dense_loss_factored", 1)
    (a, wa), (b, wb) = caps
    assert np.allclose(a['loss'], b['loss'], rtol=1e-5, atol=1e-7), (a['loss'], b['loss'])
    gmax = max(np.abs(g).max() for g in b['grads'].values() if g is not None)
    for k, gb in b['grads'].items():
        if gb is None:
            continue
        assert np.abs(a['grads'][k] - gb).max() <= 2e-5 * gmax, "%s: %g (gmax %g)" % (k, np.abs(a['grads'][k] - gb).max(), gmax)
    for k in wa:
        assert np.allclose(wa[k], wb[k], rtol=1e-3, atol=2e-3), k


@pytest.mark.parametrize("loss", ["rmse_dense", "separation_dense"])
def test_dense_losses_fit_where_the_prediction_matrix_cannot_exist(loss):
    """1,000,000 users x 100,000 items: the [n_users, n_items] prediction the reference's dense losses reduce over is 400 GB --
    more than the device holds.  The factored form fits in a few GB, the loss is finite and falls over the epochs."""
    n_users, n_items, d = 1_000_000, 100_000, 32
    rng = np.random.default_rng(0)
    cols = rng.integers(0, n_items, size=(n_users, 5), dtype=np.int32)
    inter = sp.csr_matrix((np.ones(n_users * 5, np.float32), cols.reshape(-1), np.arange(0, (n_users + 1) * 5, 5, dtype=np.int64)),
                          shape=(n_users, n_items))
    inter.sum_duplicates()
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    model = T.TensorRec(n_components=d, loss_graph=LOSS[loss](), seed=0)
    losses = []
    for _ in range(4):
        model._capture = {}
        model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05)
        losses.append(float(np.asarray(model._capture['loss']).reshape(-1)[0]))
    model._capture = None
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert torch.cuda.max_memory_allocated() < 16 * 2 ** 30, torch.cuda.max_memory_allocated()


def test_fused_wmrb_falls_back_when_rows_do_not_fit():
    """A user with more interactions than LDS can hold next to the samples -> the composed path runs (same results as
    ever); custom WMRB subclasses are never fused."""
    from tensorrec_amd import ops
    from tensorrec_amd.sparse import Interactions
    m = sp.csr_matrix(np.ones((3, 2600), np.float32))
    inter = Interactions(m, 3, 2600, "cuda")
    assert inter.max_row_nnz == 2600 and not ops.wmrb_fused_supported(40, inter, 64)
    small = Interactions(sp.csr_matrix(np.ones((3, 20), np.float32)), 3, 2600, "cuda")
    assert ops.wmrb_fused_supported(100, small, 128) and not ops.wmrb_fused_supported(300, small, 128)
    assert not ops.wmrb_fused_supported(100, small, 130)

    class MyWMRB(WMRBLossGraph):
        calls = 0

        def weighted_margin_rank_batch(self, **kw):
            MyWMRB.calls += 1
            return WMRBLossGraph.weighted_margin_rank_batch(self, **kw)

    inter, uf, itf = dummy(40, 60, seed=1)
    T.TensorRec(n_components=8, loss_graph=MyWMRB(), seed=0).fit(inter, uf, itf, epochs=2, n_sampled_items=5)
    assert MyWMRB.calls == 2


@pytest.mark.parametrize("loss,kw", [("rmse", {}), ("wmrb", {"n_sampled_items": 12}), ("balanced_wmrb", {"n_sampled_items": 12}),
                                     ("rmse_dense", {})])
@pytest.mark.parametrize("batch", [None, 25])
def test_hip_graph_replay_equals_eager_steps(loss, kw, batch):
    """A fit whose steps 2..n are HIP-graph replays (forward + backward captured after the first eager step; sampler and
    Adam outside the graph) ends with the weights of the all-eager fit: bit-identical for the deterministic RMSE steps,
    up to the summation order of the counting-sort buckets for the sampled losses."""
    inter, uf, itf = dummy(60, 90, seed=3)
    out = []
    for graphs in (True, False):
        model = T.TensorRec(n_components=16, loss_graph=LOSS[loss](), seed=9, hip_graphs=graphs)
        model.fit(inter, uf, itf, epochs=6, learning_rate=0.05, user_batch_size=batch, **kw)
        out.append((model.get_weights(), model._opt_step, model._sample_step))
    (wa, oa, sa), (wb, ob, sb) = out
    assert (oa, sa) == (ob, sb)
    for k in wa:
        if loss.startswith("rmse"):
            assert np.array_equal(wa[k], wb[k]), k
        elif k != "user_feature_biases":
            assert np.allclose(wa[k], wb[k], rtol=1e-3, atol=2e-3), k


@pytest.mark.parametrize("loss,biased,sampler,d,S", [("wmrb", True, "device", 64, 40), ("balanced_wmrb", False, "replay", 32, 25),
                                                     ("wmrb", True, "replay", 128, 60), ("wmrb", True, "device", 20, 7)])
def test_single_kernel_step_equals_multi_launch_steps(loss, biased, sampler, d, S, monkeypatch):
    """csrc/step_coop.hip: the whole optimiser step of a model that fits on chip (tensorrec.py:617-622 -- sampling, both towers,
    serial + sampled predictions, WMRB, autodiff, Adam) as ONE cooperative kernel, against the same fit made of separate launches:
    the same step and sample counters, the same weights up to summation order -- with the samples drawn INSIDE the kernel (device
    sampler: the bits of trec_sample_items, or the fits would diverge within a step) and with replayed tables; item features with
    side columns, a user without interactions, negative interactions."""
    import tensorrec_amd.tensorrec as TT
    rng = np.random.default_rng(11)
    n_users, n_items, steps = 150, 333, 7
    inter = sp.random(n_users, n_items, density=0.08, random_state=3, format="csr", dtype=np.float32)
    inter.data[:] = np.where(rng.random(inter.nnz) < 0.85, 1.0, -1.0)
    inter[7, :] = 0
    inter.eliminate_zeros()
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    import bench_records as BR
    itf = BR._side_features(n_items, 19, 3, rng)
    srng = np.random.RandomState(5)
    tables = [O.sample_items(n_items, n_users, S, False, srng)[:, 1].reshape(n_users, S) for _ in range(steps)]
    calls = {"run": 0}
    orig_run = TT._CoopStep.run

    def run(self, *a, **k):
        out = orig_run(self, *a, **k)
        calls["run"] += out is not False
        return out
    monkeypatch.setattr(TT._CoopStep, "run", run)
    out = []
    for coop in (1, 0):
        T._native.set_tuning("fit_step_coop", coop)
        try:
            kw = {"sampler": T.ReplaySampler(tables)} if sampler == "replay" else {}
            model = T.TensorRec(n_components=d, loss_graph=LOSS[loss](), biased=biased, seed=9, hip_graphs=False, **kw)
            model.fit(inter, uf, itf, epochs=1, learning_rate=0.05, n_sampled_items=S)              # eager: variables + slots
            m1 = {n: mv[0].cpu().numpy().copy() for n, mv in model._adam.items()}
            model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, n_sampled_items=S)      # the step under test
            # the step's gradient (L2 term included) read back from Adam's first moment: m2 = b1 m1 + (1 - b1) g
            g2 = {n: (mv[0].cpu().numpy() - np.float32(0.9) * m1[n]) / np.float32(0.1) for n, mv in model._adam.items()}
            model.fit_partial(inter, uf, itf, epochs=steps - 2, learning_rate=0.05, n_sampled_items=S)
            out.append((g2, model._opt_step, model._sample_step, model.predict(uf, itf)))
        finally:
            T._native.set_tuning("fit_step_coop", 1)
        if coop:
            assert calls["run"] == steps - 1, calls                  # (every step but the first)
    (ga, oa, sa, pa), (gb, ob, sb, pb) = out
    assert (oa, sa) == (ob, sb) == (steps, steps) and calls["run"] == steps - 1
    gmax = max(np.abs(g).max() for g in gb.values())
    for k in gb:
        # (read back through m: its fp32 rounding, 6e-8 |m| / (1 - b1), is part of the bar.)  Per TENSOR as well: Adam rescales
        # every variable by its own gradient history, so a bias gradient wrong by an amount that is small next to the largest
        # weight gradient still moves the fit (found that way: the bias variables' L2 term)
        err = np.abs(ga[k] - gb[k]).max()
        assert err <= 3e-5 * gmax, "%s: %g (gmax %g)" % (k, err, gmax)
        if k != "user_feature_biases":
            assert err <= 2e-3 * np.abs(gb[k]).max(), "%s: %g of %g" % (k, err, np.abs(gb[k]).max())
    # five more steps each way: Adam turns the rounding noise of near-zero gradients into steps of up to lr -- the user biases above
    # all, whose WMRB gradient is exactly 0 in exact arithmetic (b_u cancels inside every hinge) -- so the fits are compared by what
    # they predict, up to each user's constant offset (which no ranking and no WMRB loss sees)
    diff = pa - pb
    diff = diff - diff.mean(axis=1, keepdims=True)
    assert np.abs(diff).max() <= 2e-2 * max(1.0, np.abs(pb).max())


def test_hip_graph_is_actually_used(monkeypatch):
    import tensorrec_amd.tensorrec as TT
    calls = {"capture": 0, "run": 0}
    orig_capture, orig_run = TT._GraphedStep.capture.__func__, TT._GraphedStep.run

    def capture(cls, *a, **k):
        calls["capture"] += 1
        return orig_capture(cls, *a, **k)

    def run(self, *a, **k):
        calls["run"] += 1
        return orig_run(self, *a, **k)

    monkeypatch.setattr(TT._GraphedStep, "capture", classmethod(capture))
    monkeypatch.setattr(TT._GraphedStep, "run", run)
    inter, uf, itf = dummy(60, 90, seed=3)
    T.TensorRec(n_components=16, loss_graph=WMRBLossGraph(), seed=1).fit(inter, uf, itf, epochs=8, n_sampled_items=10)
    assert calls == {"capture": 1, "run": 7}
    calls.update(capture=0, run=0)
    T.TensorRec(n_components=16, seed=1, hip_graphs=False).fit(inter, uf, itf, epochs=8)
    assert calls == {"capture": 0, "run": 0}


def test_uploads_are_reused_between_fit_calls_until_the_content_changes():
    """fit_partial in a loop (the reference's evaluate-every-epoch idiom): the device copies of unchanged matrices are
    reused, a changed matrix is uploaded again, and the weights equal those of a model that never caches."""
    inter, uf, itf = T.util.generate_dummy_data(num_users=60, num_items=90, interaction_density=.2, random_state=1)
    inter, uf, itf = sp.csr_matrix(inter), sp.csr_matrix(uf), sp.csr_matrix(itf)
    a = T.TensorRec(n_components=8, seed=3)         # RMSE: every gather is deterministic
    b = T.TensorRec(n_components=8, seed=3)         # RMSE: every gather is deterministic
    b.cache_uploads = False
    for m in (a, b):
        m.fit_partial(inter, uf, itf, epochs=1)
    first = {id(v) for v in a._upload_cache.values()}
    assert len(first) == 3 and not b._upload_cache
    for m in (a, b):
        m.fit_partial(inter, uf, itf, epochs=2)
    assert {id(v) for v in a._upload_cache.values()} == first              # nothing was rebuilt
    changed = inter.copy()
    changed.data[::3] *= -1.0                                              # same structure, other values
    for m in (a, b):
        m.fit_partial(changed, uf, itf, epochs=1)
    now = {id(v) for v in a._upload_cache.values()}
    assert len(now & first) == 2 and len(now) == 3                          # the features stayed, the interactions did not
    for x, y in zip(a.predict(uf, itf), b.predict(uf, itf)):
        assert np.array_equal(x, y)


def test_rank_of_interactions_with_many_positives_per_user_sorts_rows():
    """>= 32 positives per user: the tile's rows are sorted once (K4 sorted form) and the pairs' ranks read off -- the
    same integers as the counting path and as predict_rank()."""
    _, uf, itf = dummy(90, 400, seed=4)
    inter = sp.random(90, 400, density=0.2, random_state=5, dtype=np.float32, format="csr")
    inter.data[:] = 1.0
    inter.sort_indices()
    model = T.TensorRec(n_components=16, seed=2)
    model.fit(inter, uf, itf, epochs=2)
    ranks = model.predict_rank(uf, itf)
    coo = inter.tocoo()
    for batch in (None, 32):
        pr = model.predict_rank_of_interactions(uf, itf, inter, user_batch_size=batch)
        assert len(pr.ranks) >= 32 * 90 and np.array_equal(pr.ranks, ranks[pr.rows, sp.csr_matrix(inter).tocoo().col])
    assert np.array_equal(np.sort(coo.row, kind="stable"), pr.rows)


def test_wide_models_and_many_tastes_take_the_fallbacks():
    """fit accepted n_components > 256 and n_tastes > 16 but predict* raised.  Wider models score through the
    K-looped fp32 GEMM, more tastes collapse in groups of 16 (max) or through the composed softmax form (attention);
    results against the oracle model from the same weights, top-k / ranks consistent with predict."""
    inter, uf, itf = dummy(40, 70, seed=8)
    for d, n_tastes, pred in ((300, 1, "dot"), (260, 1, "euclidean"), (24, 18, "dot")):
        oracle = OracleTensorRec(d, "linear", "linear", pred, "rmse", True, n_tastes=n_tastes)
        oracle.init_weights(uf.shape[1], itf.shape[1], np.random.default_rng(3))
        model = T.TensorRec(n_components=d, n_tastes=n_tastes, prediction_graph=PRED[pred](), seed=1)
        model.build(uf.shape[1], itf.shape[1])
        w = dict(oracle.weights)
        if n_tastes == 1:
            w = _rename(w)
        model.set_weights(w)
        p_gpu, p_ref = model.predict(uf, itf), oracle.predict(uf, itf)
        assert np.abs(p_gpu - p_ref).max() <= 1e-4 * np.abs(p_ref).max()
        assert np.array_equal(model.predict_rank(uf, itf), O.rank_predictions_exact(p_gpu))
        vals, idx = model.predict_top_k(uf, itf, k=5)
        rv, ri = O.topk_rows(p_gpu, 5)
        assert np.array_equal(idx, ri) and np.array_equal(vals, rv)
        model.fit_partial(inter, uf, itf, epochs=1)                  # and training still runs


@pytest.mark.parametrize("loss", ["wmrb", "balanced_wmrb"])
def test_deterministic_fit_is_bit_reproducible(loss):
    """TensorRec(deterministic=True): the sampled pairs are grouped by a stable sort, every fp32 sum of the step is added
    in a fixed order -- two fits from the same seed end with bit-identical weights (the default path orders a bucket by
    atomic arrival and agrees only to summation order)."""
    rng = np.random.default_rng(0)
    n_users, n_items = 3000, 700                                # ~430 sampled pairs per item: buckets with real contention
    cols = rng.integers(0, n_items, size=(n_users, 12))
    inter = sp.csr_matrix((np.ones(n_users * 12, np.float32), cols.reshape(-1), np.arange(0, n_users * 12 + 1, 12)),
                          shape=(n_users, n_items))
    inter.sum_duplicates()
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    runs = []
    for _ in range(2):
        model = T.TensorRec(n_components=32, loss_graph=LOSS[loss](), seed=5, deterministic=True)
        model.fit(inter, uf, itf, epochs=4, n_sampled_items=100)
        runs.append(model.get_weights())
    for k in runs[0]:
        assert np.array_equal(runs[0][k], runs[1][k]), k
    ref = T.TensorRec(n_components=32, loss_graph=LOSS[loss](), seed=5)        # default path: same fit to rounding
    ref.fit(inter, uf, itf, epochs=4, n_sampled_items=100)
    for k, v in ref.get_weights().items():
        if k == "user_feature_biases":                          # zero gradient in exact arithmetic: moves by noise (see above)
            continue
        # (4 Adam steps at lr 0.1: a hinge on its kink may switch between the two summation orders and move a row by O(lr))
        close = np.abs(v - runs[0][k]) <= 1e-3 * (1.0 + np.abs(runs[0][k]))
        assert close.mean() >= 0.99 and np.abs(v - runs[0][k]).max() <= 0.8, k
