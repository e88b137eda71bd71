"""
Variables for eagerly executed graphs.

In the reference the three ABC methods run ONCE, while the TF graph is built (tensorrec/tensorrec.py:270-492), and
``tf.Variable(...)`` inside them creates the trainable state.  Here the same methods run every step on device
tensors, so variable creation has to be idempotent: inside a model's ``variable_scope`` a ``Variable(init, name=..)``
call returns the existing tensor of that name, or creates it (on the model's device) the first time.

    class TanhRepresentationGraph(AbstractRepresentationGraph):             # cf. test/test_readme.py:37-66
        def connect_representation_graph(self, tf_features, n_components, n_features, node_name_ending):
            w = Variable(lambda: random_normal([n_features, n_components], stddev=.5),
                         name='tanh_weights_%s' % node_name_ending)
            return torch.tanh(sparse_tensor_dense_matmul(tf_features, w)), [w]
"""
from __future__ import annotations

import threading

import torch

_state = threading.local()


def resolve_device(device):
    """torch.device with an explicit index ('cuda' -> the current device), so that devices compare and guard reliably
    (torch.device('cuda') != torch.device('cuda:0'))."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None and torch.cuda.is_available():
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


class VariableStore(object):
    def __init__(self, device, seed=None):
        self.device = resolve_device(device)
        self.variables = {}          # name -> leaf tensor (requires_grad)
        self.order = []
        self._anon = 0
        # ONE generator per model, created once: successive initialiser draws continue one seeded stream, so equally
        # shaped weights (the tastes, equal user / item feature counts) start different.  seed=None: torch's global RNG.
        self.seed = _generator.get("seed") if seed is None else int(seed)
        self._gen = None

    def generator(self):
        if self.seed is None:
            return None
        if self._gen is None:
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(self.seed)
        return self._gen

    def get(self, name, init):
        if name is None:
            raise ValueError("Variable(name=...) is required: graph methods run every step, names keep state stable")
        if name not in self.variables:
            value = init() if callable(init) else init
            t = torch.as_tensor(value, dtype=torch.float32).detach().to(self.device).contiguous().clone()
            t.requires_grad_(True)
            self.variables[name] = t
            self.order.append(name)
        return self.variables[name]


class variable_scope(object):
    def __init__(self, store):
        self.store = store

    def __enter__(self):
        self.prev = getattr(_state, "store", None)
        _state.store = self.store
        return self.store

    def __exit__(self, *exc):
        _state.store = self.prev
        return False


def current_store():
    store = getattr(_state, "store", None)
    if store is None:
        raise RuntimeError("Variable() used outside of a TensorRec model (no variable scope is active)")
    return store


def Variable(initial_value, name=None):
    """Idempotent stand-in for ``tf.Variable(initial_value, name=name)``; ``initial_value`` may be a tensor or a
    zero-argument callable (evaluated only on first creation)."""
    return current_store().get(name, initial_value)


def device():
    return current_store().device


# initialisers with the reference's signatures -------------------------------------------------------------------
_generator = {}


def set_seed(seed):
    """Default seed of models created afterwards without one (``None`` clears it).  The reference exposes no seed
    (SURVEY.md 3.4); TensorRec(seed=...) is the per-model form and takes precedence."""
    if seed is None:
        _generator.pop("seed", None)
    else:
        _generator["seed"] = int(seed)


def random_normal(shape, stddev=1.0):
    store = current_store()
    return torch.randn(tuple(shape), dtype=torch.float32, device=store.device, generator=store.generator()) * stddev


def zeros(shape):
    return torch.zeros(tuple(shape), dtype=torch.float32, device=device())


def ones(shape):
    return torch.ones(tuple(shape), dtype=torch.float32, device=device())
