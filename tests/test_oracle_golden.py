"""The oracle against every known-answer vector the reference's tests hold for the hot path
(SURVEY.md 8c).  Vectors: tests/golden/reference_goldens.json, extracted from
/root/reference/test/*.py by tests/golden/extract_reference_goldens.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O


@pytest.mark.parametrize("kind", ["dot", "cosine", "euclidean"])
def test_prediction_graph_dense(goldens, kind):
    g = goldens[kind + "_dense"]
    got = O.DENSE[kind](g["array_1"], g["array_2"])
    # The reference asserts np.allclose (test/test_prediction_graphs.py:55,109,166) on float64
    # inputs, i.e. TF ran these graphs in float64.  The hot path is float32 (input_utils.py:16),
    # so the replay is held to float32 resolution: atol 1e-6 instead of allclose's 1e-8.
    assert np.allclose(got, g["expected_result"], atol=1e-6)


@pytest.mark.parametrize("kind", ["dot", "cosine", "euclidean"])
def test_prediction_graph_serial(goldens, kind):
    g = goldens[kind + "_serial"]
    got = O.SERIAL[kind](g["array_1"], g["array_2"], g["x_user"].astype(int), g["x_item"].astype(int))
    assert np.allclose(got, g["expected_result"], atol=1e-6)


def test_exact_variants_agree_with_goldens(goldens):
    g = goldens["dot_dense"]
    assert np.array_equal(O.score_dense_exact(g["array_1"], g["array_2"]), g["expected_result"].astype(np.float32))
    g = goldens["dot_serial"]
    got = O.pair_dot_exact(g["array_1"], g["array_2"], g["x_user"], g["x_item"])
    assert np.array_equal(got, g["expected_result"].astype(np.float32))
    g = goldens["euclidean_dense"]
    u, v = g["array_1"].astype(np.float32), g["array_2"].astype(np.float32)
    got = O.score_dense_euclid_exact(u, v, (u ** 2).sum(1), (v ** 2).sum(1))
    assert np.allclose(got, g["expected_result"])


def test_project_biases(goldens):
    g = goldens["project_biases"]
    got = O.project_biases(g["features"], g["feature_biases"])
    assert (got == g["expected_result"]).all()          # reference asserts exact equality (:40)
    got2 = O.spmm_exact(g["features"], g["feature_biases"]).sum(axis=1)
    assert (got2 == g["expected_result"]).all()


def test_split_sparse_tensor_indices(goldens):
    g = goldens["split_sparse_tensor_indices"]
    xu, xi = O.split_sparse_tensor_indices(g["interactions"])
    assert (xu == g["expected_user"]).all() and (xi == g["expected_item"]).all()
    # CSR input must serialise in the same (row-major) order
    xu2, xi2 = O.split_sparse_tensor_indices(sp.csr_matrix(g["interactions"]))
    assert (xu2 == g["expected_user"]).all() and (xi2 == g["expected_item"]).all()


def test_bias_prediction_dense(goldens):
    g = goldens["bias_prediction_dense"]
    got = O.bias_prediction_dense(g["predictions"], g["projected_user_biases"], g["projected_item_biases"])
    assert (got == g["expected_biased_predictions"]).all()


def test_bias_prediction_serial(goldens):
    g = goldens["bias_prediction_serial"]
    got = O.bias_prediction_serial(g["predictions"], g["projected_user_biases"], g["projected_item_biases"],
                                   g["x_user"].astype(int), g["x_item"].astype(int))
    assert (got == g["expected_biased_predictions"]).all()


def test_densify_sampled_item_predictions(goldens):
    g = goldens["densify_sampled_item_predictions"]
    got = O.densify_sampled_item_predictions(g["input_data"], 4, 3)
    assert (got == g["expected_result"]).all()


def test_rank_predictions(goldens):
    g = goldens["rank_predictions"]
    for fn in (O.rank_predictions, O.rank_predictions_exact, O.rank_by_counting):
        got = fn(g["predictions"])
        assert got.dtype == np.int32
        assert (got == g["expected_ranks"]).all(), fn.__name__


def test_rank_is_counting_on_heavy_ties():
    """the double top_k equals 1 + #{j: s_j > s_i or (s_j == s_i and j < i)}."""
    rng = np.random.default_rng(0)
    pred = rng.integers(0, 4, size=(7, 33)).astype(np.float32)     # many exact ties
    a, b, c = O.rank_predictions(pred), O.rank_predictions_exact(pred), O.rank_by_counting(pred)
    assert (a == b).all() and (a == c).all()
    assert (np.sort(a, axis=1) == np.arange(1, 34)[None, :]).all()    # each row is a permutation


def test_topk_rows_is_prefix_of_rank_order():
    rng = np.random.default_rng(1)
    pred = rng.integers(0, 5, size=(5, 40)).astype(np.float32)
    vals, idx = O.topk_rows(pred, 7)
    ranks = O.rank_predictions(pred)
    for u in range(5):
        assert (ranks[u, idx[u]] == np.arange(1, 8)).all()
        assert (vals[u] == pred[u, idx[u]]).all()
    vals, idx = O.topk_rows(pred[:, :3], 5)                         # fewer items than k
    assert (idx[:, 3:] == -1).all() and np.isneginf(vals[:, 3:]).all()


def test_collapse_mixture_of_tastes(goldens):
    g = goldens["collapse_mixture_of_tastes"]
    got = O.collapse_mixture_of_tastes(g["predictions"])
    assert (got == g["expected_predictions"]).all()


def test_collapse_mixture_of_tastes_with_attention(goldens):
    g = goldens["collapse_mixture_of_tastes_with_attention"]
    got = O.collapse_mixture_of_tastes(g["predictions"], g["attentions"])
    # the reference asserts float32 equality against TF's softmax kernel; a NumPy exp differs from
    # Eigen's in the last ulp, so the oracle is held to 2 ulp here (n_tastes > 1 is a NEXT row, 8f)
    assert np.allclose(got, g["expected_predictions"], rtol=3e-7, atol=0)


def test_predict_similar_items(goldens):
    g = goldens["predict_similar_items"]
    got = O.predict_similar_items(O.cosine_dense, g["reprs"], [1])
    assert (got == g["expected_sims"]).all()


def test_calculate_batched_alpha(goldens):
    g = goldens["calculate_batched_alpha"]
    for case in g["cases"]:
        got = O.calculate_batched_alpha(num_batches=case["num_batches"], alpha=case["alpha"])
        if case["places"] is None:
            assert got == case["expected"]
        else:
            assert round(abs(got - case["expected"]), case["places"]) == 0
    with pytest.raises(ValueError):
        O.calculate_batched_alpha(num_batches=g["raises_value_error_for_num_batches"], alpha=.01)


def test_exact_and_blas_flavours_agree():
    rng = np.random.default_rng(2)
    u = rng.standard_normal((17, 24)).astype(np.float32)
    v = rng.standard_normal((29, 24)).astype(np.float32)
    ub, ib = rng.standard_normal(17).astype(np.float32), rng.standard_normal(29).astype(np.float32)
    a = O.score_dense_exact(u, v, ub, ib)
    b = O.bias_prediction_dense(O.dot_dense(u, v), ub, ib)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
    x = sp.random(13, 40, density=0.2, random_state=3, dtype=np.float32, format="csr")
    w = rng.standard_normal((40, 12)).astype(np.float32)
    assert np.allclose(O.spmm_exact(x, w), O.spmm(x, w), rtol=1e-5, atol=1e-6)
