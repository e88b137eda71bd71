"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (NumPy / SciPy / torch-CPU float32) of the arithmetic on TensorRec's
scoring + training hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under
``tensorrec_amd/`` does (tests/test_host_logic.py::test_no_oracle_in_product enforces it).

Why a restatement: the reference (jfkirk/tensorrec v0.26.2, /root/reference) is pure
Python over TensorFlow 1.x (``tensorflow>=1.7.0``, setup.py:21).  TensorFlow is not
vendored and not installable here (no wheel, no network), so the package cannot be
imported.  Each function below cites the reference file:line it follows; TF op
semantics that are not in the reference tree are marked [external].

Pinning status (also in DESIGN.md):
  * PINNED by the reference's own known-answer tests (tests/golden/reference_goldens.json,
    extracted mechanically from /root/reference/test/*.py by
    tests/golden/extract_reference_goldens.py): dot / cosine / euclidean prediction
    graphs (dense + serial), project_biases, split_sparse_tensor_indices,
    bias_prediction_dense/serial, densify_sampled_item_predictions, rank_predictions,
    collapse_mixture_of_tastes (max + softmax attention), predict_similar_items,
    calculate_batched_alpha.
  * PARITY UNPINNED by the reference (its tests only smoke-run these): representation
    graph numerics, RMSE / WMRB / BalancedWMRB loss values, sample_items output, the
    Adam step and any end-to-end fit result.  For those the formulas below are the only
    pin; they were additionally cross-checked by executing the reference's own
    ``loss_graphs.py`` / ``representation_graphs.py`` source on a NumPy stand-in for the
    TF ops it calls (tests/golden/run_reference_on_shim.py -> reference_shim_goldens.json).

Two numeric flavours:
  * ``*_exact`` functions call oracle/_build/libtr_oracle.so (tr_oracle.c): fmaf
    chains in a fixed order, comparable bit-for-bit with the fp32 HIP kernels.
  * the plain functions use NumPy/torch float32 (BLAS order), used for tolerance
    checks, gradients (torch-CPU autograd stands in for TF's autodiff) and as the
    timed CPU baseline.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtr_oracle.so")
_lib = None


def build_c_oracle(force: bool = False) -> str:
    """Compile tr_oracle.c with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libtr_oracle.so"])
    return _LIB_PATH


def _c():
    global _lib
    if _lib is None:
        build_c_oracle()
        lib = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        lib.orc_spmm_csr.argtypes = [vp, vp, vp, vp, i64, vp, i32, vp]
        lib.orc_score_dense.argtypes = [vp, vp, i64, i64, i32, vp, vp, vp]
        lib.orc_score_dense_euclid.argtypes = [vp, vp, i64, i64, i32, vp, vp, vp, vp, vp]
        lib.orc_rank_rows.argtypes = [vp, i64, i64, vp]
        lib.orc_topk_rows.argtypes = [vp, i64, i64, i32, vp, vp]
        lib.orc_pair_dot.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp]
        for f in ("orc_spmm_csr", "orc_score_dense", "orc_score_dense_euclid", "orc_rank_rows",
                  "orc_topk_rows", "orc_pair_dot"):
            getattr(lib, f).restype = None
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- #
# input plumbing                                                              #
# --------------------------------------------------------------------------- #
def to_coo_like_reference(m):
    """input_utils.py:29-36: ``sp.coo_matrix(m)``; rows/cols int64, values float32.
    For CSR input SciPy emits row-major order, which is the order of
    ``tf_interactions.values`` / ``tf_prediction_serial`` everywhere downstream."""
    if not isinstance(m, sp.coo_matrix):
        m = sp.coo_matrix(m)
    return m.row.astype(np.int64), m.col.astype(np.int64), m.data.astype(np.float32), m.shape


def csr_arrays(m):
    m = sp.csr_matrix(m)
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float32)


# --------------------------------------------------------------------------- #
# TF primitives [external semantics]                                          #
# --------------------------------------------------------------------------- #
def l2_normalize_rows(x, epsilon=1e-12):
    """tf.nn.l2_normalize(x, 1) [external]: x * rsqrt(max(sum(x**2, 1), epsilon))."""
    x = _f32(x)
    ss = np.sum(x * x, axis=1, keepdims=True, dtype=np.float32)
    return x * (np.float32(1.0) / np.sqrt(np.maximum(ss, np.float32(epsilon))))


def spmm(features, w):
    """tf.sparse_tensor_dense_matmul (representation_graphs.py:40,119;
    recommendation_graphs.py:15) with SciPy CSR @ dense, float32."""
    return _f32(sp.csr_matrix(features).astype(np.float32) @ _f32(w))


def spmm_exact(features, w):
    indptr, indices, vals = csr_arrays(features)
    w = _f32(w)
    out = np.empty((len(indptr) - 1, w.shape[1]), np.float32)
    _c().orc_spmm_csr(_p(indptr), _p(indices), _p(vals), None, len(indptr) - 1, _p(w), w.shape[1], _p(out))
    return out


# --------------------------------------------------------------------------- #
# representation graphs -- tensorrec/representation_graphs.py                 #
# --------------------------------------------------------------------------- #
def linear_repr(features, w):                      # :32-43
    return spmm(features, w)


def normalized_linear_repr(features, w):           # :53-58
    return l2_normalize_rows(spmm(features, w))


def feature_passthrough_repr(features, n_components):   # :66-74
    features = sp.csr_matrix(features)
    if n_components != features.shape[1]:
        raise ValueError("FeaturePassThroughRepresentationGraph requires n_features and n_components to be equal.")
    return _f32(features.toarray())


def weighted_feature_passthrough_repr(features, n_components, weights=None):   # :82-89
    dense = feature_passthrough_repr(features, n_components)
    if weights is None:
        weights = np.ones((1, n_components), np.float32)   # tf.ones, a constant (:87)
    return dense * _f32(weights)


def relu_repr(features, w_relu, b_relu, w_linear):  # :102-124
    h = np.maximum(spmm(features, w_relu) + _f32(b_relu), np.float32(0.0))
    return _f32(h @ _f32(w_linear))


def init_linear_weights(n_features, n_components, rng):
    """:35-36 random_normal(stddev=1) then l2_normalize(axis=1), at init only."""
    return l2_normalize_rows(rng.standard_normal((n_features, n_components)).astype(np.float32))


def init_relu_weights(n_features, n_components, rng, relu_size=None):
    """:105-116 -- stddev .5 normals, zero biases, relu_size = 4*n_components."""
    relu_size = 4 * n_components if relu_size is None else relu_size
    return ((0.5 * rng.standard_normal((n_features, relu_size))).astype(np.float32),
            np.zeros((1, relu_size), np.float32),
            (0.5 * rng.standard_normal((relu_size, n_components))).astype(np.float32))


# --------------------------------------------------------------------------- #
# prediction graphs -- tensorrec/prediction_graphs.py                         #
# --------------------------------------------------------------------------- #
def dot_dense(u, v):                               # :49-50
    return _f32(_f32(u) @ _f32(v).T)


def dot_serial(u, v, xu, xi):                      # :52-55
    return np.sum(_f32(u)[xu] * _f32(v)[xi], axis=1, dtype=np.float32)


def relative_cosine(t1, t2):                       # recommendation_graphs.py:112-121
    return _f32(l2_normalize_rows(t1) @ l2_normalize_rows(t2).T)


def cosine_dense(u, v):                            # :64-65
    return relative_cosine(u, v)


def cosine_serial(u, v, xu, xi):                   # :67-72
    return np.sum(l2_normalize_rows(u)[xu] * l2_normalize_rows(v)[xi], axis=1, dtype=np.float32)


EUCLID_EPS = np.float32(1e-16)                     # :82


def euclid_dense(u, v):                            # :84-100
    u, v = _f32(u), _f32(v)
    r_u = np.sum(u ** 2, 1, keepdims=True, dtype=np.float32)
    r_v = np.sum(v ** 2, 1, keepdims=True, dtype=np.float32)
    dist = (r_u - np.float32(2.0) * (u @ v.T)) + r_v.T
    return np.float32(-1.0) * np.sqrt(np.maximum(dist, EUCLID_EPS))


def euclid_serial(u, v, xu, xi):                   # :102-117
    delta = (_f32(u)[xu] - _f32(v)[xi]) ** 2
    dist = np.maximum(np.sum(delta, axis=1, dtype=np.float32), EUCLID_EPS)
    return np.float32(-1.0) * np.sqrt(dist)


DENSE = {"dot": dot_dense, "cosine": cosine_dense, "euclidean": euclid_dense}
SERIAL = {"dot": dot_serial, "cosine": cosine_serial, "euclidean": euclid_serial}


def score_dense_exact(u, v, user_bias=None, item_bias=None):
    """fmaf-chain dot product (+ biases in the reference's order); tr_oracle.c."""
    u, v = _f32(u), _f32(v)
    out = np.empty((u.shape[0], v.shape[0]), np.float32)
    ub = None if user_bias is None else _f32(user_bias)
    ib = None if item_bias is None else _f32(item_bias)
    _c().orc_score_dense(_p(u), _p(v), u.shape[0], v.shape[0], u.shape[1], _p(ub), _p(ib), _p(out))
    return out


def score_dense_euclid_exact(u, v, r_user, r_item, user_bias=None, item_bias=None):
    u, v = _f32(u), _f32(v)
    out = np.empty((u.shape[0], v.shape[0]), np.float32)
    ub = None if user_bias is None else _f32(user_bias)
    ib = None if item_bias is None else _f32(item_bias)
    ru, ri = _f32(r_user), _f32(r_item)
    _c().orc_score_dense_euclid(_p(u), _p(v), u.shape[0], v.shape[0], u.shape[1], _p(ru), _p(ri),
                                _p(ub), _p(ib), _p(out))
    return out


def pair_dot_exact(u, v, xu, xi, user_bias=None, item_bias=None):
    u, v = _f32(u), _f32(v)
    xu = np.ascontiguousarray(xu, np.int32)
    xi = np.ascontiguousarray(xi, np.int32)
    out = np.empty(len(xu), np.float32)
    ub = None if user_bias is None else _f32(user_bias)
    ib = None if item_bias is None else _f32(item_bias)
    _c().orc_pair_dot(_p(u), _p(v), _p(xu), _p(xi), len(xu), u.shape[1], _p(ub), _p(ib), _p(out))
    return out


# --------------------------------------------------------------------------- #
# recommendation graphs -- tensorrec/recommendation_graphs.py                 #
# --------------------------------------------------------------------------- #
def project_biases(features, feature_biases):      # :4-19
    return np.sum(spmm(features, _f32(feature_biases).reshape(-1, 1)), axis=1, dtype=np.float32)


def split_sparse_tensor_indices(m):                # :22-30
    rows, cols, _, _ = to_coo_like_reference(m)
    return rows, cols


def bias_prediction_dense(pred, ub, ib):           # :33-41
    return _f32(pred) + _f32(ub)[:, None] + _f32(ib)[None, :]


def bias_prediction_serial(pred, ub, ib, xu, xi):  # :44-57
    return _f32(pred) + _f32(ub)[xu] + _f32(ib)[xi]


def densify_sampled_item_predictions(serial, n_sampled_items, n_users):   # :60-70
    return np.reshape(serial, (int(n_users), int(n_sampled_items)))


def rank_predictions(pred):                        # :73-82
    """Literal restatement of the double top_k.  tf.nn.top_k is descending and puts the
    lower index first among equal values [external]; a stable argsort of -pred is that."""
    pred = _f32(pred)
    n_items = pred.shape[1]
    indices_of_ranks = np.argsort(-pred, axis=1, kind="stable")          # top_k(pred)[1]
    ranks = np.argsort(indices_of_ranks, axis=1, kind="stable")          # top_k(-idx)[1]
    del n_items
    return (ranks + 1).astype(np.int32)


def rank_predictions_exact(pred):
    pred = _f32(pred)
    out = np.empty(pred.shape, np.int32)
    _c().orc_rank_rows(_p(pred), pred.shape[0], pred.shape[1], _p(out))
    return out


def rank_by_counting(pred):
    """The identity the HIP rank kernel uses:
    rank_i = 1 + #{j : s_j > s_i or (s_j == s_i and j < i)}.  O(I^2); small inputs only."""
    pred = _f32(pred)
    n = pred.shape[1]
    j = np.arange(n)
    gt = pred[:, None, :] > pred[:, :, None]                    # [u, i, j]: s_j > s_i
    eq_lower = (pred[:, None, :] == pred[:, :, None]) & (j[None, None, :] < j[None, :, None])
    return (1 + np.sum(gt | eq_lower, axis=2)).astype(np.int32)


def topk_rows(pred, k):
    """First k columns of the first top_k in rank_predictions: (value desc, index asc)."""
    pred = _f32(pred)
    vals = np.empty((pred.shape[0], k), np.float32)
    idx = np.empty((pred.shape[0], k), np.int32)
    _c().orc_topk_rows(_p(pred), pred.shape[0], pred.shape[1], k, _p(vals), _p(idx))
    return vals, idx


def collapse_mixture_of_tastes(tastes_predictions, tastes_attentions=None):   # :85-109
    stacked = np.stack([_f32(p) for p in tastes_predictions])
    if tastes_attentions is not None:
        att = np.stack([_f32(a) for a in tastes_attentions])
        att = att - np.max(att, axis=0, keepdims=True)          # tf.nn.softmax is max-shifted [external]
        e = np.exp(att)
        soft = e / np.sum(e, axis=0, keepdims=True, dtype=np.float32)
        return np.sum(stacked * soft, axis=0, dtype=np.float32)
    return np.max(stacked, axis=0)


def predict_similar_items(dense_fn, item_repr, item_ids):      # :124-137
    item_repr = _f32(item_repr)
    return dense_fn(item_repr[np.asarray(item_ids)], item_repr)


# --------------------------------------------------------------------------- #
# loss graphs -- tensorrec/loss_graphs.py                                     #
# --------------------------------------------------------------------------- #
def rmse_loss(pred_serial, interactions_serial):   # :58-59
    e = _f32(interactions_serial) - _f32(pred_serial)
    return np.sqrt(np.mean(e * e, dtype=np.float32))


def rmse_dense_loss(prediction, interactions):   # :62-72  tf.sparse_add(interactions, -prediction), mean over U*I
    err = -1.0 * _f32(prediction)
    m = sp.coo_matrix(interactions)
    np.add.at(err, (m.row, m.col), m.data.astype(np.float32))
    return np.sqrt(np.mean(err * err, dtype=np.float32))


def _separation(pos, neg):                       # :90-96  tf.nn.moments = population variance; Normal(loc, scale).cdf(0)
    pos, neg = _f32(pos), _f32(neg)
    pos_mean, neg_mean = pos.mean(dtype=np.float32), neg.mean(dtype=np.float32)
    pos_var = np.mean((pos - pos_mean) ** 2, dtype=np.float32)
    neg_var = np.mean((neg - neg_mean) ** 2, dtype=np.float32)
    loc = neg_mean - pos_mean
    scale = np.sqrt(neg_var + pos_var)
    cdf0 = 0.5 * (1.0 + math.erf(float((0.0 - loc) / (scale * np.float32(math.sqrt(2.0))))))
    return np.float32(1.0 - cdf0)


def separation_loss(pred_serial, interactions_serial):   # :75-97
    pred_serial, y = _f32(pred_serial), _f32(interactions_serial)
    return _separation(pred_serial[y > 0.0], pred_serial[y <= 0.0])


def separation_dense_loss(prediction, interactions):     # :100-134: non-interacted pairs count as negatives
    dense = np.asarray(sp.coo_matrix(interactions).todense(), dtype=np.float32).reshape(-1)
    pred = _f32(prediction).reshape(-1)
    return _separation(pred[dense > 0.0], pred[dense <= 0.0])


def wmrb_loss(pred_serial, x_user, values, sample_predictions, n_items, n_sampled_items):   # :153-180
    """Returns the [P+] vector (the reference does NOT reduce it, :179-180)."""
    values = _f32(values)
    mask = values > 0.0
    pos_pred = _f32(pred_serial)[mask]
    mapped = _f32(sample_predictions)[np.asarray(x_user)[mask]]            # gather by USER index (:167-168)
    summation = np.maximum(np.float32(1.0) - pos_pred[:, None] + mapped, np.float32(0.0))
    ratio = np.float32(n_items) / np.float32(n_sampled_items)
    smr = ratio * np.sum(summation, axis=1, dtype=np.float32)
    return np.log(smr + np.float32(1.0))


def balanced_wmrb_loss(pred_serial, x_user, x_item, values, sample_predictions, n_items, n_sampled_items,
                       interactions_shape):        # :189-227
    values = _f32(values)
    mask = values > 0.0
    pos_vals = values[mask]
    pos_items = np.asarray(x_item)[mask]
    per_item = np.zeros(interactions_shape[1], np.float32)
    np.add.at(per_item, pos_items, pos_vals)                               # sparse_reduce_sum(axis=0)
    gathered = per_item[pos_items]
    pos_pred = _f32(pred_serial)[mask]
    mapped = _f32(sample_predictions)[np.asarray(x_user)[mask]]
    summation = np.maximum(np.float32(1.0) - pos_pred[:, None] + mapped, np.float32(0.0))
    ratio = np.float32(n_items) / np.float32(n_sampled_items)
    smr = ratio * np.sum(summation, axis=1, dtype=np.float32) * pos_vals / gathered
    return np.log(smr + np.float32(1.0))


# --------------------------------------------------------------------------- #
# util.py                                                                     #
# --------------------------------------------------------------------------- #
def sample_items(n_items, n_users, n_sampled_items, replace, rng=np.random):   # util.py:12-21
    """One ``choice`` per user, pairs emitted user-major; nothing excludes positives."""
    items_per_user = [rng.choice(a=n_items, size=n_sampled_items, replace=replace) for _ in range(n_users)]
    sample_indices = []
    for user, users_items in enumerate(items_per_user):
        for item in users_items:
            sample_indices.append((user, item))
    return np.array(sample_indices, np.int64).reshape(-1, 2)


def calculate_batched_alpha(num_batches, alpha):   # util.py:24-31
    if num_batches < 1:
        raise ValueError("num_batches must be >=1, num_batches={}".format(num_batches))
    elif num_batches > 1:
        return alpha / (math.e * math.log(num_batches))
    return alpha


# --------------------------------------------------------------------------- #
# optimiser: tf.train.AdamOptimizer [external: TF 1.x training_ops ApplyAdam]   #
# --------------------------------------------------------------------------- #
ADAM_B1, ADAM_B2, ADAM_EPS = np.float32(0.9), np.float32(0.999), np.float32(1e-8)


def adam_powers(t):
    """beta1_power / beta2_power after t steps: TF keeps them as float32 variables that are
    multiplied by beta once per step, starting from beta itself [external]."""
    b1p, b2p = np.float32(1.0), np.float32(1.0)
    for _ in range(int(t)):
        b1p = np.float32(b1p * ADAM_B1)
        b2p = np.float32(b2p * ADAM_B2)
    return b1p, b2p


def adam_lr_t(lr, t):
    """lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) evaluated in float32 (t is 1-based)."""
    b1p, b2p = adam_powers(t)
    return np.float32(np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p))


def adam_tf_step(w, m, v, g, lr_t):
    """In-place on float32 arrays, element order of the TF CPU functor:
        m += (g - m) * (1 - b1);  v += (g*g - v) * (1 - b2);  w -= (m * lr_t) / (sqrt(v) + eps)
    Every op is a separately rounded float32 op (NumPy does not fuse)."""
    one = np.float32(1.0)
    m += (g - m) * (one - ADAM_B1)
    v += (g * g - v) * (one - ADAM_B2)
    w -= (m * np.float32(lr_t)) / (np.sqrt(v) + ADAM_EPS)
    return w, m, v
