"""The oracle against fixtures produced by EXECUTING THE REFERENCE'S OWN graph source on a NumPy stand-in for TensorFlow
(tests/golden/run_reference_on_shim.py -> reference_shim_goldens.json).  These cover what the reference's tests leave
unpinned: representation graphs and every loss value.  Tolerances are float32 round-off of different
summation orders; integer outputs are exact."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_goldens
from oracle import oracle as O


@pytest.fixture(scope="module")
def sg():
    return load_goldens("reference_shim_goldens.json")


TOL = dict(rtol=1e-5, atol=1e-6)


def test_representation_graphs(sg):
    g = sg["repr_linear"]
    w = g["variables"]["linear_weights_user"]
    assert np.allclose(np.linalg.norm(w, axis=1), 1.0, atol=1e-6)        # weights are row-normalised at init (:35-36)
    assert np.allclose(O.linear_repr(g["features"], w), g["expected_repr"], **TOL)
    assert np.allclose(O.spmm_exact(g["features"], w), g["expected_repr"], **TOL)
    g = sg["repr_normalized_linear"]
    assert np.allclose(O.normalized_linear_repr(g["features"], g["variables"]["linear_weights_user"]),
                       g["expected_repr"], **TOL)
    for key, relu_size in (("repr_relu", 32), ("repr_relu_size_5", 5)):
        g = sg[key]
        v = g["variables"]
        assert v["relu_weights_user"].shape == (17, relu_size) and (v["relu_biases_user"] == 0).all()
        # the reference returns [relu_weights, linear_weights, relu_biases] (representation_graphs.py:124)
        assert g["weights_order"] == ["relu_weights_user", "linear_weights_user", "relu_biases_user"]
        got = O.relu_repr(g["features"], v["relu_weights_user"], v["relu_biases_user"], v["linear_weights_user"])
        assert np.allclose(got, g["expected_repr"], rtol=1e-5, atol=1e-5)
    g = sg["repr_passthrough"]
    assert g["n_weights"] == 0 and np.array_equal(O.feature_passthrough_repr(g["features"], 8), g["expected_repr"])
    g = sg["repr_weighted_passthrough"]
    assert g["n_weights"] == 1
    assert np.array_equal(O.weighted_feature_passthrough_repr(g["features"], 8), g["expected_repr"])


def test_losses(sg):
    g = sg["loss_rmse"]
    rows, cols, vals, shape = O.to_coo_like_reference(sp.csr_matrix(g["interactions"]))
    assert np.allclose(O.rmse_loss(g["prediction_serial"], vals), g["expected_loss"], **TOL)
    g = sg["loss_wmrb"]
    got = O.wmrb_loss(g["prediction_serial"], rows, vals, g["sample_predictions"], g["n_items"], g["n_sampled_items"])
    assert got.shape == g["expected_loss"].shape == (int((vals > 0).sum()),)
    assert np.allclose(got, g["expected_loss"], **TOL)
    g = sg["loss_balanced_wmrb"]
    got = O.balanced_wmrb_loss(g["prediction_serial"], rows, cols, vals, g["sample_predictions"], g["n_items"],
                               g["n_sampled_items"], shape)
    assert np.allclose(got, g["expected_loss"], **TOL)
    # dense + separation losses (loss_graphs.py:62-134)
    g = sg["loss_separation"]
    assert np.allclose(O.separation_loss(g["prediction_serial"], vals), g["expected_loss"], **TOL)
    for key, fn in (("loss_rmse_dense", O.rmse_dense_loss), ("loss_separation_dense", O.separation_dense_loss)):
        g = sg[key]
        dense = O.DENSE["dot"](g["user_repr"], g["item_repr"])
        assert np.allclose(fn(dense, sp.csr_matrix(g["interactions"])), g["expected_loss"], **TOL), key


def test_prediction_graphs_and_ranks(sg):
    for kind in ("dot", "cosine", "euclidean"):
        g = sg["pred_" + kind]
        xu, xi = g["x_user"].astype(int), g["x_item"].astype(int)
        assert np.allclose(O.DENSE[kind](g["user_repr"], g["item_repr"]), g["expected_dense"], rtol=1e-5, atol=1e-5)
        assert np.allclose(O.SERIAL[kind](g["user_repr"], g["item_repr"], xu, xi), g["expected_serial"], rtol=1e-5,
                           atol=1e-5)
    g = sg["rank_predictions_ties"]
    for fn in (O.rank_predictions, O.rank_predictions_exact, O.rank_by_counting):
        assert np.array_equal(fn(g["predictions"]), g["expected_ranks"].astype(np.int32))


def test_oracle_model_matches_reference_losses(sg):
    """oracle.model's torch restatement of the WMRB wiring reproduces the reference's loss vector."""
    import torch
    g = sg["loss_wmrb"]
    rows, cols, vals, shape = O.to_coo_like_reference(sp.csr_matrix(g["interactions"]))
    pred, samp = torch.from_numpy(g["prediction_serial"]), torch.from_numpy(g["sample_predictions"])
    mask = torch.from_numpy(vals > 0)
    summ = torch.clamp(1.0 - pred[mask][:, None] + samp[torch.from_numpy(rows)[mask]], min=0.0)
    loss = torch.log((float(g["n_items"]) / float(g["n_sampled_items"])) * summ.sum(1) + 1.0)
    assert np.allclose(loss.numpy(), g["expected_loss"], **TOL)
