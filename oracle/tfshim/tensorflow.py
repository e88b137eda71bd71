"""
oracle/tfshim/tensorflow.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy stand-in for the handful of TensorFlow-1.x symbols that the reference's L2 graph modules call
(tensorrec/representation_graphs.py, prediction_graphs.py, recommendation_graphs.py, loss_graphs.py), eager instead
of graph-building.  It exists so that tests/golden/run_reference_on_shim.py can EXECUTE THE REFERENCE'S OWN SOURCE
(read from /root/reference, never copied) on seeded inputs and record its outputs as fixtures.

What such fixtures pin: the reference's composition of ops (which tensors are gathered, masked, broadcast, reduced,
in which order).  What they do not pin: TensorFlow's kernels themselves -- every op below is this file's NumPy
reading of the documented TF semantics ([external]), float32 throughout.  TensorFlow itself is not installable here.
"""
import math as _math
import types as _types

import numpy as np

float32, int32, int64 = np.float32, np.int32, np.int64
__version__ = "1.13.1-numpy-shim"
_rng = np.random.RandomState(0)


def set_random_seed(seed):
    global _rng
    _rng = np.random.RandomState(seed)


def _a(x):
    if isinstance(x, SparseTensor):
        return x
    return np.asarray(x, dtype=np.float32) if np.asarray(x).dtype.kind == "f" else np.asarray(x)


class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
        self.values = np.asarray(values, dtype=np.float32)
        self.dense_shape = np.asarray(dense_shape, dtype=np.int64)


VARIABLES = {}


def Variable(initial_value, name=None):
    v = np.array(initial_value, dtype=np.float32)
    VARIABLES[name if name is not None else "var_%d" % len(VARIABLES)] = v
    return v


def random_normal(shape, stddev=1.0):
    return (_rng.standard_normal(tuple(int(s) for s in shape)) * stddev).astype(np.float32)


def zeros(shape):
    return np.zeros(tuple(int(s) for s in shape), np.float32)


def ones(shape):
    return np.ones(tuple(int(s) for s in shape), np.float32)


def sparse_tensor_dense_matmul(sp_a, b):
    b = _a(b)
    out = np.zeros((int(sp_a.dense_shape[0]), b.shape[1]), np.float32)
    for (r, c), v in zip(sp_a.indices, sp_a.values):          # COO order, one multiply-add per entry
        out[r] += np.float32(v) * b[c]
    return out


def sparse_tensor_to_dense(sp_a, validate_indices=True):
    out = np.zeros(tuple(int(s) for s in sp_a.dense_shape), np.float32)
    for (r, c), v in zip(sp_a.indices, sp_a.values):
        out[r, c] = v
    return out


def sparse_reduce_sum(sp_a, axis=None):
    return np.sum(sparse_tensor_to_dense(sp_a), axis=axis, dtype=np.float32)


def sparse_add(a, b):
    return sparse_tensor_to_dense(a) + _a(b)


def multiply(a, b):
    return _a(a) * _a(b)


def add(a, b):
    return _a(a) + _a(b)


def matmul(a, b, transpose_b=False):
    a, b = _a(a), _a(b)
    return (a @ (b.T if transpose_b else b)).astype(np.float32)


def gather(params, indices):
    return _a(params)[np.asarray(indices, dtype=np.int64)]


def reduce_sum(x, axis=None, keep_dims=False):
    return np.sum(_a(x), axis=axis, keepdims=keep_dims, dtype=np.float32)


def reduce_mean(x, axis=None):
    return np.mean(_a(x), axis=axis, dtype=np.float32)


def reduce_max(x, axis=None):
    return np.max(_a(x), axis=axis)


def square(x):
    return _a(x) * _a(x)


def pow(x, y):  # noqa: A001
    return np.power(_a(x), np.float32(y))


def maximum(x, y):
    return np.maximum(_a(x), np.float32(y) if np.isscalar(y) else _a(y))


def sqrt(x):
    return np.sqrt(_a(x))


def log(x):
    return np.log(_a(x))


def transpose(x):
    return np.transpose(np.asarray(x))


def expand_dims(x, axis):
    return np.expand_dims(_a(x), axis)


def stack(values, axis=0):
    return np.stack([np.asarray(v) for v in values], axis=axis)


def shape(x):
    return np.array(x.dense_shape if isinstance(x, SparseTensor) else np.asarray(x).shape, dtype=np.int32)


def reshape(x, shape):  # noqa: A002
    return np.reshape(np.asarray(x), tuple(int(s) for s in np.asarray(shape).reshape(-1)))


def cast(x, dtype):
    return np.asarray(x).astype(dtype)


def greater(x, y):
    return _a(x) > y


def less_equal(x, y):
    return _a(x) <= y


def boolean_mask(tensor, mask):
    return np.asarray(tensor)[np.asarray(mask, dtype=bool)]


def _l2_normalize(x, axis, epsilon=1e-12):
    x = _a(x)
    ss = np.sum(x * x, axis=axis, keepdims=True, dtype=np.float32)
    return x * (np.float32(1.0) / np.sqrt(np.maximum(ss, np.float32(epsilon))))


def _relu(x):
    return np.maximum(_a(x), np.float32(0.0))


def _softmax(x, axis=-1):
    x = _a(x)
    e = np.exp(x - np.max(x, axis=axis, keepdims=True))
    return e / np.sum(e, axis=axis, keepdims=True, dtype=np.float32)


def _top_k(x, k):
    x = _a(x)
    order = np.argsort(-x, axis=-1, kind="stable")[..., : int(k)]        # descending, lower index first on ties
    return np.take_along_axis(x, order, axis=-1), order.astype(np.int32)


def _moments(x, axes):
    x = _a(x)
    mean = np.mean(x, axis=tuple(axes), dtype=np.float32)
    var = np.mean((x - mean) ** 2, axis=tuple(axes), dtype=np.float32)     # population variance
    return mean, var


def _l2_loss(x):
    return np.float32(0.5) * np.sum(_a(x) ** 2, dtype=np.float32)


class _Normal(object):
    def __init__(self, loc, scale):
        self.loc, self.scale = np.float32(loc), np.float32(scale)

    def cdf(self, x):
        return np.float32(0.5 * (1.0 + _math.erf((float(x) - float(self.loc)) / (float(self.scale) * _math.sqrt(2.0)))))


nn = _types.SimpleNamespace(l2_normalize=_l2_normalize, relu=_relu, softmax=_softmax, top_k=_top_k, moments=_moments,
                            l2_loss=_l2_loss)
contrib = _types.SimpleNamespace(distributions=_types.SimpleNamespace(Normal=_Normal))
