#!/usr/bin/env python3
"""
Generate tests/golden/reference_shim_goldens.json by EXECUTING THE REFERENCE'S OWN SOURCE for the four L2 graph
modules on seeded inputs.  TensorFlow is absent, so the modules are imported with oracle/tfshim/tensorflow.py (a NumPy
reading of the TF ops they call) standing in for `tensorflow`.  The reference files are loaded from /root/reference at
run time (package `tensorrec` is stubbed so that its __init__ -- which pulls in sessions, tf.data, ... -- never runs);
nothing is copied into this repository except the numbers produced.

What this pins (and what not): see oracle/tfshim/tensorflow.py.  These fixtures cover exactly the rows the reference's
own tests leave unpinned: representation-graph outputs, RMSE / WMRB / BalancedWMRB loss values.

    python tests/golden/run_reference_on_shim.py          # needs /root/reference; the JSON is committed
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TENSORREC_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def load_reference_modules():
    from oracle.tfshim import tensorflow as tf
    sys.modules["tensorflow"] = tf
    pkg = types.ModuleType("tensorrec")
    pkg.__path__ = [os.path.join(REF, "tensorrec")]          # submodules resolve here; tensorrec/__init__.py is NOT run
    sys.modules["tensorrec"] = pkg
    mods = {}
    for name in ("recommendation_graphs", "prediction_graphs", "representation_graphs", "loss_graphs"):
        mods[name] = importlib.import_module("tensorrec." + name)
    return tf, mods


def arr(a):
    a = np.asarray(a)
    return {"__array__": True, "dtype": str(a.dtype), "data": a.astype(np.float64).tolist()}


def sparse(m):
    m = sp.coo_matrix(m)
    return {"__sparse__": True, "shape": list(m.shape), "row": m.row.tolist(), "col": m.col.tolist(),
            "data": m.data.astype(np.float64).tolist()}


def to_tf_sparse(tf, m):
    m = sp.coo_matrix(sp.csr_matrix(m))                        # row-major order, like input_utils.py:29-36 on CSR input
    return tf.SparseTensor(np.stack([m.row, m.col], axis=1), m.data.astype(np.float32), m.shape)


def main():
    tf, mods = load_reference_modules()
    R, P, REC, L = (mods["representation_graphs"], mods["prediction_graphs"], mods["recommendation_graphs"],
                    mods["loss_graphs"])
    rng = np.random.RandomState(7)
    out = {"_source": "reference source executed on oracle/tfshim (NumPy stand-in for TF); see run_reference_on_shim.py"}

    # ---- representation graphs (representation_graphs.py:26-124): weights are whatever the reference code creates
    n, f, d = 23, 17, 8
    feats = sp.random(n, f, density=0.3, random_state=1, dtype=np.float32, format="csr")
    feats[3, :] = 0
    feats = sp.csr_matrix(feats)
    feats.eliminate_zeros()
    for key, graph in (("linear", R.LinearRepresentationGraph()), ("normalized_linear", R.NormalizedLinearRepresentationGraph()),
                       ("relu", R.ReLURepresentationGraph()), ("relu_size_5", R.ReLURepresentationGraph(relu_size=5))):
        tf.set_random_seed(11)
        tf.VARIABLES.clear()
        repr_, weights = graph.connect_representation_graph(to_tf_sparse(tf, feats), d, f, "user")
        out["repr_" + key] = {"reference": "tensorrec/representation_graphs.py (%s)" % type(graph).__name__,
                              "features": sparse(feats), "n_components": d,
                              "variables": {k: arr(v) for k, v in tf.VARIABLES.items()},
                              "weights_order": [next(k for k, v in tf.VARIABLES.items() if v is w) for w in weights],
                              "expected_repr": arr(repr_)}
    dense_feats = sp.random(9, d, density=0.5, random_state=2, dtype=np.float32, format="csr")
    for key, graph in (("passthrough", R.FeaturePassThroughRepresentationGraph()),
                       ("weighted_passthrough", R.WeightedFeaturePassThroughRepresentationGraph())):
        repr_, weights = graph.connect_representation_graph(to_tf_sparse(tf, dense_feats), d, d, "item")
        out["repr_" + key] = {"reference": "tensorrec/representation_graphs.py (%s)" % type(graph).__name__,
                              "features": sparse(dense_feats), "n_components": d, "n_weights": len(weights),
                              "expected_repr": arr(repr_)}

    # ---- losses (loss_graphs.py:53-59, :137-227) on a small interaction matrix with positive and negative entries
    n_users, n_items, S = 11, 19, 6
    inter = sp.random(n_users, n_items, density=0.25, random_state=3, dtype=np.float32, format="csr")
    inter.data = (inter.data - 0.35).astype(np.float32)
    inter[4, :] = 0
    inter = sp.csr_matrix(inter)
    inter.eliminate_zeros()
    tf_inter = to_tf_sparse(tf, inter)
    pred_serial = rng.standard_normal(inter.nnz).astype(np.float32)
    sample_pred = rng.standard_normal((n_users, S)).astype(np.float32)
    common = {"interactions": sparse(inter), "prediction_serial": arr(pred_serial), "sample_predictions": arr(sample_pred),
              "n_items": n_items, "n_sampled_items": S}
    kwargs = dict(tf_prediction_serial=pred_serial, tf_interactions_serial=tf_inter.values, tf_interactions=tf_inter,
                  tf_n_users=n_users, tf_n_items=n_items, tf_sample_predictions=sample_pred, tf_n_sampled_items=S,
                  tf_prediction=None, tf_rankings=None)
    for key, graph in (("rmse", L.RMSELossGraph()), ("wmrb", L.WMRBLossGraph()), ("balanced_wmrb", L.BalancedWMRBLossGraph()),
                       ("separation", L.SeparationLossGraph())):
        out["loss_" + key] = dict(common, reference="tensorrec/loss_graphs.py (%s)" % type(graph).__name__,
                                  expected_loss=arr(graph.connect_loss_graph(**kwargs)))
    # dense losses need the dense prediction
    u = rng.standard_normal((n_users, 5)).astype(np.float32)
    v = rng.standard_normal((n_items, 5)).astype(np.float32)
    dense = P.DotProductPredictionGraph().connect_dense_prediction_graph(u, v)
    kwargs["tf_prediction"] = dense
    for key, graph in (("rmse_dense", L.RMSEDenseLossGraph()), ("separation_dense", L.SeparationDenseLossGraph())):
        out["loss_" + key] = {"reference": "tensorrec/loss_graphs.py (%s)" % type(graph).__name__,
                              "interactions": sparse(inter), "user_repr": arr(u), "item_repr": arr(v),
                              "expected_loss": arr(graph.connect_loss_graph(**kwargs))}

    # ---- prediction graphs on random (non-toy) inputs, dense + serial, incl. the euclidean dense/serial asymmetry
    xu = rng.randint(0, n_users, 40)
    xi = rng.randint(0, n_items, 40)
    for key, graph in (("dot", P.DotProductPredictionGraph()), ("cosine", P.CosineSimilarityPredictionGraph()),
                       ("euclidean", P.EuclideanSimilarityPredictionGraph())):
        out["pred_" + key] = {"reference": "tensorrec/prediction_graphs.py (%s)" % type(graph).__name__,
                              "user_repr": arr(u), "item_repr": arr(v), "x_user": arr(xu), "x_item": arr(xi),
                              "expected_dense": arr(graph.connect_dense_prediction_graph(u, v)),
                              "expected_serial": arr(graph.connect_serial_prediction_graph(u, v, xu, xi))}
    # ---- rank_predictions on random scores with ties (recommendation_graphs.py:73-82)
    scores = rng.randint(0, 6, (7, 31)).astype(np.float32)
    out["rank_predictions_ties"] = {"reference": "tensorrec/recommendation_graphs.py:73-82", "predictions": arr(scores),
                                    "expected_ranks": arr(REC.rank_predictions(scores))}

    path = os.path.join(HERE, "reference_shim_goldens.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path, "with", len(out) - 1, "cases")


if __name__ == "__main__":
    main()
