#!/usr/bin/env python3
"""
Extract the reference's own known-answer vectors into tests/golden/reference_goldens.json.

The reference package cannot be imported here (it needs TensorFlow 1.x, which is absent),
so instead of re-typing its expected values this script parses the reference's test files
with ``ast`` and evaluates ONLY the literal array assignments inside the named test
methods (``np.array([...])``, ``sp.coo_matrix([...])``, lists of arrays, ``math.sqrt``).
Nothing from the reference is executed beyond those literals, and no reference source is
copied into this repository -- only the numbers its tests assert.

Run inside the authoring container (needs /root/reference):
    python tests/golden/extract_reference_goldens.py
The JSON it writes is committed; the GPU box never needs /root/reference.
"""
import ast
import json
import math
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

REF = os.environ.get("TENSORREC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")

# (key, file, class, method, variables to pull)
CASES = [
    ("dot_dense", "test/test_prediction_graphs.py", "DotProductTestCase", "test_dense_prediction",
     ["array_1", "array_2", "expected_result"]),
    ("dot_serial", "test/test_prediction_graphs.py", "DotProductTestCase", "test_serial_prediction",
     ["array_1", "array_2", "x_user", "x_item", "expected_result"]),
    ("cosine_dense", "test/test_prediction_graphs.py", "CosineSimilarityTestCase", "test_dense_prediction",
     ["array_1", "array_2", "expected_result"]),
    ("cosine_serial", "test/test_prediction_graphs.py", "CosineSimilarityTestCase", "test_serial_prediction",
     ["array_1", "array_2", "x_user", "x_item", "expected_result"]),
    ("euclidean_dense", "test/test_prediction_graphs.py", "EuclideanSimilarityTestCase", "test_dense_prediction",
     ["array_1", "array_2", "expected_result"]),
    ("euclidean_serial", "test/test_prediction_graphs.py", "EuclideanSimilarityTestCase", "test_serial_prediction",
     ["array_1", "array_2", "x_user", "x_item", "expected_result"]),
    ("project_biases", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase", "test_project_biases",
     ["features", "n_features", "expected_result"]),
    ("split_sparse_tensor_indices", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_split_sparse_tensor_indices", ["interactions", "expected_user", "expected_item"]),
    ("bias_prediction_dense", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_bias_prediction_dense",
     ["predictions", "projected_user_biases", "projected_item_biases", "expected_biased_predictions"]),
    ("bias_prediction_serial", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_bias_prediction_serial",
     ["predictions", "projected_user_biases", "projected_item_biases", "x_user", "x_item",
      "expected_biased_predictions"]),
    ("densify_sampled_item_predictions", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_densify_sampled_item_predictions", ["input_data", "expected_result"]),
    ("rank_predictions", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_rank_predictions", ["predictions", "expected_ranks"]),
    ("collapse_mixture_of_tastes", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_collapse_mixture_of_tastes", ["predictions", "expected_predictions"]),
    ("collapse_mixture_of_tastes_with_attention", "test/test_recommendation_graphs.py",
     "RecommendationGraphsTestCase", "test_collapse_mixture_of_tastes_with_attention",
     ["predictions", "attentions", "expected_predictions"]),
    ("predict_similar_items", "test/test_recommendation_graphs.py", "RecommendationGraphsTestCase",
     "test_predict_similar_items", ["reprs", "expected_sims"]),
]

# numpy names the reference's tests use; np.int / np.mat were removed from NumPy 2.x
_np = types.SimpleNamespace(array=np.array, float32=np.float32, int=int, mat=np.asmatrix)
_NS = {"np": _np, "math": math, "sp": sp, "__builtins__": {}}


def _find_method(tree, cls, method):
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == method:
                    return fn
    raise KeyError((cls, method))


def _jsonable(v):
    if sp.issparse(v):
        m = sp.coo_matrix(v)
        return {"__sparse__": True, "shape": list(m.shape), "row": m.row.tolist(), "col": m.col.tolist(),
                "data": m.data.astype(np.float64).tolist()}
    if isinstance(v, np.ndarray):
        return {"__array__": True, "dtype": str(v.dtype), "data": v.astype(np.float64).tolist()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (int, float)):
        return v
    raise TypeError(type(v))


def main():
    out = {"_source": "jfkirk/tensorrec v0.26.2 test suite; extracted by tests/golden/extract_reference_goldens.py"}
    for key, rel, cls, method, wanted in CASES:
        path = os.path.join(REF, rel)
        with open(path) as fh:
            tree = ast.parse(fh.read())
        fn = _find_method(tree, cls, method)
        found = {}
        for node in ast.walk(fn):
            if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
                name = node.targets[0].id
                if name in wanted and name not in found:
                    value = eval(compile(ast.Expression(node.value), path, "eval"), dict(_NS))
                    found[name] = (_jsonable(value), node.lineno)
        missing = [w for w in wanted if w not in found]
        if missing:
            raise SystemExit("missing %s in %s::%s.%s" % (missing, rel, cls, method))
        out[key] = {"reference": "%s:%d-%d (%s.%s)" % (rel, fn.lineno, fn.end_lineno, cls, method)}
        for name, (val, _) in found.items():
            out[key][name] = val

    # project_biases assigns the variable values inline (test_recommendation_graphs.py:35)
    out["project_biases"]["feature_biases"] = {"__array__": True, "dtype": "float32",
                                               "data": [[-.5], [.5], [0], [2.0]]}
    # calculate_batched_alpha, test/test_util.py:8-24 -- scalar assertions, not arrays
    out["calculate_batched_alpha"] = {"reference": "test/test_util.py:8-24",
                                      "cases": [{"num_batches": 1, "alpha": .01, "expected": .01, "places": None},
                                                {"num_batches": 2, "alpha": .01, "expected": .53074 * .01,
                                                 "places": 5}],
                                      "raises_value_error_for_num_batches": 0}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", OUT, "with", len(out) - 1, "cases")


if __name__ == "__main__":
    sys.exit(main())
