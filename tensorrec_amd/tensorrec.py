"""
TensorRec -- the public object API of tensorrec/tensorrec.py:26-917, executed on one MI355X.

Same constructor, ``fit`` / ``fit_partial`` / ``predict`` / ``predict_rank`` / ``predict_similar_items`` /
``predict_*_representation`` / ``predict_*_bias`` signatures, the same validation errors and NumPy return types.
What differs is underneath: there is no TF graph or session; the three graph objects are executed eagerly every step
on device tensors (``_forward`` mirrors the wiring of ``_build_tf_graph``, tensorrec.py:270-492), gradients come from
hand-written backward kernels chained by torch autograd, and the optimiser is one fused TF-form Adam kernel per
weight tensor.  Extensions (keyword-only, after the reference's arguments): ``precision``, ``device``, ``sampler``,
``seed`` and the ``predict_top_k`` method (fused MFMA score + top-k, no [U, I] matrix).

Parity-critical behaviours reproduced on purpose: vector losses are summed together with a
broadcast ``alpha * reg`` (so L2 is scaled by the loss length); Adam is dense over every weight; samples are shared
per user and may contain positives; the interaction matrix takes its shape from the feature matrices.
"""
from __future__ import annotations

import logging
import math
import os
import pickle
from itertools import cycle

import numpy as np
import torch
from scipy import sparse as sp

from . import ops
from . import _native as N
from .errors import (
    ModelNotBiasedException, ModelNotFitException, ModelWithoutAttentionException, BatchNonSparseInputException
)
from .framework import VariableStore, variable_scope, resolve_device
from .loss_graphs import (AbstractLossGraph, RMSELossGraph, RMSEDenseLossGraph, WMRBLossGraph,
                          BalancedWMRBLossGraph, SeparationLossGraph, SeparationDenseLossGraph)
from .prediction_graphs import AbstractPredictionGraph, DotProductPredictionGraph
from .recommendation_graphs import (
    project_biases, bias_prediction_dense, bias_prediction_serial, rank_predictions,
    densify_sampled_item_predictions, predict_similar_items
)
from .representation_graphs import AbstractRepresentationGraph, LinearRepresentationGraph
from .sparse import SparseFeatures, Interactions, PairIndex
from .util import calculate_batched_alpha, sample_items

ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON = 0.9, 0.999, 1e-8      # tf.train.AdamOptimizer defaults [external]
PREDICT_CACHE_MAX_NNZ = 200_000_000     # non-zeros of feature matrices the predict* calls keep uploaded (~4 GB of device memory)


class DeviceSampler(object):
    """K7: counter-based sampling on the GPU (keyed permutation per user when ``replace`` is False)."""

    def __init__(self, seed=0):
        self.seed = int(seed)

    def sample(self, n_items, n_users, n_sampled_items, replace, step, device, user_base=0):
        return ops.sample_items(n_users, n_items, n_sampled_items, replace, self.seed, step, device, user_base)


class HostSampler(object):
    """The reference's sampler verbatim in behaviour (util.sample_items on the host, util.py:12-21), uploaded each
    step.  ``rng`` is anything with ``.choice`` (default: the global ``np.random`` as in the reference)."""

    def __init__(self, rng=None):
        self.rng = rng if rng is not None else np.random

    def __getstate__(self):
        # the default rng is the np.random MODULE, which cannot be pickled (save_model pickles the model object)
        return {"rng": None if self.rng is np.random else self.rng}

    def __setstate__(self, state):
        self.rng = state["rng"] if state.get("rng") is not None else np.random

    def sample(self, n_items, n_users, n_sampled_items, replace, step, device, user_base=0):
        pairs = sample_items(n_items, n_users, n_sampled_items, replace, rng=self.rng)
        items = pairs[:, 1].reshape(n_users, n_sampled_items).astype(np.int32)
        return torch.from_numpy(items).to(device)


class ReplaySampler(object):
    """Consumes host-supplied [n_users, n_sampled_items] index tables, one per step (exact parity runs)."""

    def __init__(self, tables):
        self.tables = list(tables)
        self.pos = 0

    def sample(self, n_items, n_users, n_sampled_items, replace, step, device, user_base=0):
        table = np.ascontiguousarray(self.tables[self.pos % len(self.tables)], dtype=np.int32)
        self.pos += 1
        if table.shape != (n_users, n_sampled_items):
            raise ValueError("replayed sample table has shape %s, expected %s" % (table.shape, (n_users, n_sampled_items)))
        return torch.from_numpy(table).to(device)


def _adam_lr_t(lr, t):
    """lr * sqrt(1 - b2^t) / (1 - b1^t) with the float32 running powers TF keeps [external]."""
    b1p, b2p = np.float32(1.0), np.float32(1.0)
    for _ in range(int(t)):
        b1p = np.float32(b1p * np.float32(ADAM_BETA1))
        b2p = np.float32(b2p * np.float32(ADAM_BETA2))
    return float(np.float32(np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p)))


def _to_host(t):
    """Device tensor -> NumPy array.  Large results (the [n_users, n_items] score / rank matrices of ``predict`` and
    ``predict_rank``) are copied into page-locked memory, where the DMA engines run at PCIe rate instead of staging
    through the runtime's bounce buffer (~7 GB/s from pageable memory); the array keeps the pinned block alive and
    torch's host allocator recycles it when the caller drops the array."""
    nbytes = t.numel() * t.element_size()
    if not t.is_cuda or nbytes < (16 << 20) or nbytes > (8 << 30):
        return t.cpu().numpy()
    t = t.contiguous()
    try:
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:            # pragma: no cover  (no page-locked memory to be had)
        return t.cpu().numpy()
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def _fingerprint(matrix):
    """Content key of a CSR matrix (shape, nnz, dtype, xxh3-128 of indptr / indices / data), or None when it cannot be
    had cheaply (not CSR, or no xxhash module): such matrices are uploaded afresh."""
    if not sp.isspmatrix_csr(matrix):
        return None
    try:
        import xxhash
    except ImportError:          # pragma: no cover
        return None
    h = xxhash.xxh3_128()
    for a in (matrix.indptr, matrix.indices, matrix.data):
        h.update(np.ascontiguousarray(a).view(np.uint8).data)
    return (tuple(matrix.shape), int(matrix.nnz), str(matrix.data.dtype), h.hexdigest())


def _on_model_device(method):
    """Run a public method with the MODEL's GPU as the current device: every kernel is launched with raw pointers on
    ``torch.cuda.current_stream()``, which belongs to the current device -- a model on cuda:1 in a process whose current
    device is cuda:0 would otherwise launch on GPU 0 with GPU 1's pointers."""
    import functools

    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        if self._store is None and not torch.cuda.is_available():
            return method(self, *args, **kwargs)            # the method itself raises (not fit / no GPU)
        dev = self._store.device if self._store is not None else self._device()
        if dev.type != 'cuda':
            return method(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return method(self, *args, **kwargs)
    return guarded


class TensorRec(object):
    cache_uploads = True          # defaults for objects unpickled from before these attributes existed
    _upload_cache = {}
    dp_sync_every_call = True
    _dp_plan = _dp_prev_plan = _dp_batches = None
    _dp_stale_rows = _dp_stale_slots = False


    def __init__(self,
                 n_components=100,
                 n_tastes=1,
                 user_repr_graph=LinearRepresentationGraph(),
                 item_repr_graph=LinearRepresentationGraph(),
                 attention_graph=None,
                 prediction_graph=DotProductPredictionGraph(),
                 loss_graph=RMSELossGraph(),
                 biased=True,
                 *,
                 precision='fp32',
                 device=None,
                 sampler=None,
                 seed=None,
                 data_parallel=False,
                 process_group=None,
                 hip_graphs=True,
                 deterministic=False,
                 dp_sync_every_call=True):
        """
        A TensorRec recommendation model (arguments as tensorrec/tensorrec.py:28-61).
        :param precision: 'fp32' (default; exact float32 results: fp32 MFMA, or -- top-k on large catalogues -- the bf16
        MFMA contraction as an error-bounded filter with fp32 re-scoring of the survivors, same bits) or 'bf16' (bf16
        operands, fp32 accumulate, approximate) for predict / predict_rank / predict_top_k.
        :param deterministic: if True, the item-side sums of sample-based fits are added in pair order (a stable sort
        of the sampled pairs instead of the atomic counting sort): fits are bit-reproducible run to run, ~25% slower
        at 1M x 1M.  Default False: reproducible up to fp32 summation order, like TF's own GPU kernels.
        :param device: torch device; default 'cuda' (there is no CPU execution path).
        :param sampler: object with ``sample(n_items, n_users, n_sampled_items, replace, step, device)``;
        default DeviceSampler(seed).
        :param seed: int or None -- seeds weight initialisation and the default sampler.
        :param data_parallel: if True and torch.distributed is initialised, ``fit`` is data-parallel over USERS: every
        rank passes its own rows of ``interactions`` / ``user_features`` (and ``user_offset=`` its first global user
        index) plus the full ``item_features``; weight gradients are all-reduced (RCCL) once per step, so a step
        equals the single-process step on the union of the shards (the reference's ``user_batch_size=None`` case).
        Needs ``seed`` (identical initial weights on every rank) and a loss that is a per-interaction vector (the
        WMRB family); see sharding.py.
        :param dp_sync_every_call: data-parallel fit only.  Tables whose gradient rows are rank-disjoint (identity / one-hot
        user features under user sharding) are stepped by their owner alone, with NO exchange per step; the other ranks'
        copies of those rows are refreshed by one broadcast round at the END of every ``fit_partial`` call -- every rank is
        inside the call then, so a later predict / get_weights on one rank alone sees current weights.  False: the refresh is
        left to an explicit ``dp_sync()`` (a collective: every rank must call it) -- for loops of one-epoch ``fit_partial``
        calls.  The Adam slots of rows a rank does not own are only brought together when a fit call changes the row ownership,
        or by ``dp_sync(optimizer_state=True)`` (needed before ``save_model``).
        :param hip_graphs: capture the forward + backward of a small, launch-bound training step in a HIP graph after its
        first eager execution and replay it for the remaining epochs of a ``fit`` call (the sampler and the optimiser
        stay outside the graph: their step counters change every step).  Large steps run eagerly either way.
        """
        # Arg Check (tensorrec.py:68-88)
        if (n_components is None) or (n_tastes is None) or (user_repr_graph is None) or (item_repr_graph is None) \
                or (prediction_graph is None) or (loss_graph is None):
            raise ValueError("All arguments to TensorRec() must be non-None")
        if n_components < 1:
            raise ValueError("n_components must be >= 1")
        if n_tastes < 1:
            raise ValueError("n_tastes must be >= 1")
        if not isinstance(user_repr_graph, AbstractRepresentationGraph):
            raise ValueError("user_repr_graph must inherit AbstractRepresentationGraph")
        if not isinstance(item_repr_graph, AbstractRepresentationGraph):
            raise ValueError("item_repr_graph must inherit AbstractRepresentationGraph")
        if not isinstance(prediction_graph, AbstractPredictionGraph):
            raise ValueError("prediction_graph must inherit AbstractPredictionGraph")
        if not isinstance(loss_graph, AbstractLossGraph):
            raise ValueError("loss_graph must inherit AbstractLossGraph")
        if attention_graph is not None:
            if not isinstance(attention_graph, AbstractRepresentationGraph):
                raise ValueError("attention_graph must be None or inherit AbstractRepresentationGraph")
            if n_tastes == 1:
                raise ValueError("attention_graph must be None if n_tastes == 1")
        if precision not in ('fp32', 'bf16'):
            raise ValueError("precision must be 'fp32' or 'bf16'")

        self.n_components = n_components
        self.n_tastes = n_tastes
        self.user_repr_graph_factory = user_repr_graph
        self.item_repr_graph_factory = item_repr_graph
        self.attention_graph_factory = attention_graph
        self.prediction_graph_factory = prediction_graph
        self.loss_graph_factory = loss_graph
        self.biased = biased
        self.precision = precision
        self.device = device
        self.seed = seed
        self.sampler = sampler
        self.data_parallel = bool(data_parallel)
        self.process_group = process_group
        self.hip_graphs = bool(hip_graphs)
        self.deterministic = bool(deterministic)
        self.dp_sync_every_call = bool(dp_sync_every_call)
        self._dp_plan = None              # sharding.GradPlan of the latest fit call (data-parallel fit)
        self._dp_stale_rows = False       # rows of rank-disjoint tables in this replica wait for their owners' values
        self._dp_stale_slots = False      # ... and Adam slots of rows this rank does not own (needed by save_model / a new plan)
        self.cache_uploads = True          # reuse device copies of matrices whose content did not change between fit calls
        self._upload_cache = {}
        if self.data_parallel and seed is None:
            raise ValueError("data_parallel=True needs seed= so that every rank starts from the same weights")

        self._store = None
        self._graph_pool_owner = []
        self.last_route = None        # what predict_top_k did last (route report)
        self.last_step_form = None    # how the last training step ran: "coop" (one cooperative kernel), "graph" (HIP-graph replay), "eager"
        self._capture = None          # tests set this to a dict to receive the last step's loss and raw gradients
        self._adam = {}
        self._opt_step = 0
        self._sample_step = 0
        self._schedule = None         # device-side {beta powers, lr_t, sample step} for HIP-graph replays
        self._schedule_mirror = None
        self.n_user_features = None
        self.n_item_features = None

    # ------------------------------------------------------------------------------------------ helpers
    @property
    def is_fit(self):
        return self._store is not None

    def _dp_active(self):
        if not self.data_parallel:
            return False
        import torch.distributed as dist
        from . import sharding
        return sharding.active(self.process_group)

    def _device(self):
        N.require_gpu()
        N.load()
        return resolve_device(self.device if self.device is not None else 'cuda')

    @staticmethod
    def _as_list(raw_input):
        """util.datasets_from_raw_input (util.py:34-58): a scipy matrix, a dataset in the standard TensorRec format, the
        path of a TFRecord file, or a list of those.  Everything becomes scipy CSR on the host (one upload per call);
        ``tf.data.Dataset`` objects themselves are TensorFlow containers and cannot exist here."""
        from .input_utils import TensorRecDataset, create_tensorrec_dataset_from_tfrecord

        def one(v):
            if sp.issparse(v):
                return v
            if isinstance(v, TensorRecDataset):
                return v.to_sparse_matrix()
            if isinstance(v, str):
                return create_tensorrec_dataset_from_tfrecord(v).to_sparse_matrix()
            return None

        if isinstance(raw_input, list):
            mats = [one(v) for v in raw_input]
            if all(m is not None for m in mats):
                return mats
        else:
            m = one(raw_input)
            if m is not None:
                return [m]
        raise ValueError('Input must be a scipy sparse matrix, an iterable of scipy sprase matrices, or a TensorFlow '
                         'Dataset')

    def _create_batches(self, interactions, user_features, item_features, user_batch_size=None):
        """tensorrec.py:185-235 -- user batching by CSR row slices, then zip with the (cycled) item features."""
        if user_batch_size is not None:
            if (not sp.issparse(interactions)) or (not sp.issparse(user_features)):
                raise BatchNonSparseInputException()
            if not isinstance(interactions, sp.csr_matrix):
                interactions = sp.csr_matrix(interactions)
            if not isinstance(user_features, sp.csr_matrix):
                user_features = sp.csr_matrix(user_features)
            n_users = user_features.shape[0]
            interactions_batched, user_features_batched = [], []
            start_batch = 0
            while start_batch < n_users:
                end_batch = min(start_batch + user_batch_size, n_users)
                interactions_batched.append(interactions[start_batch:end_batch])
                user_features_batched.append(user_features[start_batch:end_batch])
                start_batch = end_batch
            interactions, user_features = interactions_batched, user_features_batched

        int_l, uf_l, if_l = self._as_list(interactions), self._as_list(user_features), self._as_list(item_features)
        if len(int_l) != len(uf_l):
            raise ValueError('Number of batches in user_features and interactions must be equal.')
        if (len(if_l) > 1) and (len(if_l) != len(uf_l)):
            raise ValueError('Number of batches in item_features must be 1 or equal to the number of batches in '
                             'user_features.')
        return list(zip(int_l, uf_l, cycle(if_l)))

    # ------------------------------------------------------------------------------------------ graph wiring
    def _is_engine_graph(self):
        return getattr(self.prediction_graph_factory, 'engine_mode', None) is not None

    def _multi(self):
        return self.n_tastes > 1 or self.attention_graph_factory is not None

    def _user_representations(self, user_feats):
        """One user representation per taste and, with attention, one attention representation per taste
        (tensorrec.py:339-356).  Returns (user_reprs, attention_reprs or None, weights)."""
        user_reprs, attn_reprs, weights = [], ([] if self.attention_graph_factory is not None else None), []
        for taste in range(self.n_tastes):
            user_repr, user_weights = self.user_repr_graph_factory.connect_representation_graph(
                tf_features=user_feats, n_components=self.n_components, n_features=self.n_user_features,
                node_name_ending='user_{}'.format(taste))
            user_reprs.append(user_repr)
            weights.extend(user_weights)
            if self.attention_graph_factory is not None:
                attn_repr, attn_weights = self.attention_graph_factory.connect_representation_graph(
                    tf_features=user_feats, n_components=self.n_components, n_features=self.n_user_features,
                    node_name_ending='attn_{}'.format(taste))
                attn_reprs.append(attn_repr)
                weights.extend(attn_weights)
        return user_reprs, attn_reprs, weights

    def _representations(self, user_feats, item_feats):
        """Item repr, user (and attention) reprs per taste and projected biases -- tensorrec.py:308-313, :339-356,
        :421-430.  Returns (user_reprs list, attn_reprs list or None, item_repr, user_bias, item_bias, weights)."""
        item_repr, item_weights = self.item_repr_graph_factory.connect_representation_graph(
            tf_features=item_feats, n_components=self.n_components, n_features=self.n_item_features,
            node_name_ending='item')
        user_reprs, attn_reprs, user_weights = self._user_representations(user_feats)
        weights = list(item_weights) + list(user_weights)
        user_bias = item_bias = None
        if self.biased:
            if user_feats is not None:
                ub_var, user_bias = project_biases(user_feats, self.n_user_features, name='user_feature_biases')
                weights.append(ub_var)
            if item_feats is not None:
                ib_var, item_bias = project_biases(item_feats, self.n_item_features, name='item_feature_biases')
                weights.append(ib_var)
        return user_reprs, attn_reprs, item_repr, user_bias, item_bias, weights

    def _serial_multi(self, user_ins, attn_ins, item_in, x_user, x_item, user_bias, item_bias, sampled=False):
        """Serial predictions of a mixture-of-tastes model: per-taste serial predictions (tensorrec.py:384-395), their
        attentions (:357-372), the collapse (:411-418) and the serial biases (:437-449), the last two in one kernel.

        For the SAMPLED pairs the reference builds the attention from ``tf_user_representation`` instead of
        ``tf_attention_representation`` (tensorrec.py:367-372), i.e. the sampled attentions ARE the sampled
        predictions; that wiring is reproduced (the per-taste tensors are simply used twice)."""
        graph = self.prediction_graph_factory

        def one(user_in):
            if self._is_engine_graph():
                return ops.pair_score(user_in, item_in, x_user, x_item, graph.engine_mode, None, None)
            return graph.connect_serial_prediction_graph(tf_user_representation=user_in,
                                                         tf_item_representation=item_in,
                                                         tf_x_user=x_user, tf_x_item=x_item)

        preds = [one(u) for u in user_ins]
        attns = None
        if attn_ins is not None:
            attns = preds if sampled else [one(a) for a in attn_ins]
        return ops.collapse_tastes(preds, attns, user_bias if self.biased else None,
                                   item_bias if self.biased else None, x_user, x_item)

    def _dense_multi(self, user_reprs, attn_reprs, item_repr, user_bias, item_bias, differentiable=False):
        """tf_prediction of a mixture-of-tastes model: per-taste dense graphs (tensorrec.py:357-360, :380-383), the
        collapse (:407-410) and the bias broadcast (:432-435)."""
        graph = self.prediction_graph_factory
        ub = user_bias if self.biased else None
        ib = item_bias if self.biased else None
        if self._is_engine_graph() and not differentiable:
            dtype = ops.DTYPE_BF16 if self.precision == 'bf16' else ops.DTYPE_F32
            want_sq = graph.engine_mode == ops.MODE_EUCLIDEAN
            wide = item_repr.shape[1] > ops.SCORE_KMAX
            if not wide:
                i_op, i_sq, kpad = ops.score_prep(item_repr, dtype, normalize=graph.engine_normalize, want_sqnorm=want_sq)
            n_u, n_i = user_reprs[0].shape[0], item_repr.shape[0]

            def stack(reprs):
                out = torch.empty((len(reprs), n_u, n_i), dtype=torch.float32, device=item_repr.device)
                for t, r in enumerate(reprs):
                    if wide:
                        ops.dense_scores(r, item_repr, dtype, graph.engine_normalize, graph.engine_mode, out=out[t])
                        continue
                    u_op, u_sq, _ = ops.score_prep(r, dtype, normalize=graph.engine_normalize, want_sqnorm=want_sq)
                    ops.score_store(u_op, i_op, dtype, kpad, None, None, graph.engine_mode, u_sq, i_sq, out=out[t])
                return out

            preds = stack(user_reprs)
            attns = stack(attn_reprs) if attn_reprs is not None else None
            return ops.collapse_tastes(preds, attns, ub.detach() if ub is not None else None,
                                       ib.detach() if ib is not None else None)

        def one(user_repr):
            if self._is_engine_graph():
                return _differentiable_dense(graph, user_repr, item_repr)
            return graph.connect_dense_prediction_graph(tf_user_representation=user_repr,
                                                        tf_item_representation=item_repr)

        preds = [one(u) for u in user_reprs]
        attns = [one(a) for a in attn_reprs] if attn_reprs is not None else None
        return ops.collapse_tastes(preds, attns, ub, ib)

    def _serial(self, user_repr, item_repr, x_user, x_item, user_bias, item_bias):
        """connect_serial_prediction_graph + bias_prediction_serial (tensorrec.py:384-395, :437-449)."""
        graph = self.prediction_graph_factory
        pred = graph.connect_serial_prediction_graph(tf_user_representation=user_repr,
                                                     tf_item_representation=item_repr,
                                                     tf_x_user=x_user, tf_x_item=x_item)
        if self.biased:
            pred = bias_prediction_serial(pred, user_bias, item_bias, x_user, x_item)
        return pred

    def _serial_engine(self, u_in, i_in, x_user, x_item, user_bias, item_bias):
        """Built-in prediction graphs: gather + contraction + both bias gathers in ONE kernel (K3)."""
        return ops.pair_score(u_in, i_in, x_user, x_item, self.prediction_graph_factory.engine_mode,
                              user_bias if self.biased else None, item_bias if self.biased else None)

    def _dense_prediction(self, user_repr, item_repr, user_bias, item_bias, differentiable=False):
        """tf_prediction: dense graph + bias broadcast (tensorrec.py:380-383, :432-435)."""
        graph = self.prediction_graph_factory
        if self._is_engine_graph() and not differentiable:
            dtype = ops.DTYPE_BF16 if self.precision == 'bf16' else ops.DTYPE_F32
            ub = user_bias.detach().contiguous() if self.biased else None
            ib = item_bias.detach().contiguous() if self.biased else None
            return ops.dense_scores(user_repr, item_repr, dtype, graph.engine_normalize, graph.engine_mode, ub, ib)
        if self._is_engine_graph() and differentiable:
            pred = _differentiable_dense(graph, user_repr, item_repr)
        else:
            pred = graph.connect_dense_prediction_graph(tf_user_representation=user_repr,
                                                        tf_item_representation=item_repr)
        if self.biased:
            pred = bias_prediction_dense(pred, user_bias, item_bias)
        return pred

    # ------------------------------------------------------------------------------------------ fit
    def fit(self, interactions, user_features, item_features, epochs=100, learning_rate=0.1, alpha=0.00001,
            verbose=False, user_batch_size=None, n_sampled_items=None, user_offset=0):
        """Constructs the model (first call) and fits it -- arguments as tensorrec/tensorrec.py:494-526."""
        self.fit_partial(interactions=interactions, user_features=user_features, item_features=item_features,
                         epochs=epochs, learning_rate=learning_rate, alpha=alpha, verbose=verbose,
                         user_batch_size=user_batch_size, n_sampled_items=n_sampled_items, user_offset=user_offset)

    @_on_model_device
    def fit_partial(self, interactions, user_features, item_features, epochs=1, learning_rate=0.1,
                    alpha=0.00001, verbose=False, user_batch_size=None, n_sampled_items=None, user_offset=0):
        """One or more epochs; one optimiser step per user batch (tensorrec/tensorrec.py:539-634).
        ``user_offset`` (extension): global index of this call's first user row -- keys the device sampler so that a
        user shard draws what the whole population would."""
        loss_graph = self.loss_graph_factory
        if loss_graph.is_sample_based:
            if (n_sampled_items is None) or (n_sampled_items <= 0):
                raise ValueError("n_sampled_items must be an integer >0")
        if (n_sampled_items is not None) and (not loss_graph.is_sample_based):
            logging.warning('n_sampled_items was specified, but the loss graph is not sample-based')

        if verbose:
            logging.info('Processing interaction and feature data')
        batches = self._create_batches(interactions, user_features, item_features, user_batch_size)

        device = self._device()
        if self._store is None:
            # numbers of features are learned from the first batch and cannot change (tensorrec.py:598-605)
            n_user_features = batches[0][1].shape[1]
            n_item_features = batches[0][2].shape[1]
            self.n_user_features, self.n_item_features = int(n_user_features), int(n_item_features)
            self._store = VariableStore(device, seed=self.seed)
            if self.sampler is None:
                self.sampler = DeviceSampler(self.seed if self.seed is not None else 0)

        # upload once per call; the epoch loop below touches only device memory
        # Matrices whose CONTENT was uploaded by the previous call are reused (xxh3 of the CSR arrays, ~20 ms for 20M
        # interactions): the reference's idiom `for epoch: model.fit_partial(..., epochs=1); evaluate` would otherwise
        # spend 0.3 s per call on the host-side transposition and upload of an ML-20M-sized matrix for a 15 ms epoch.
        dev_batches = []
        item_cache = {}
        used = {}

        def cached(kind, matrix, extra, build):
            key = _fingerprint(matrix) if self.cache_uploads else None
            if key is None:
                return build()
            key = (kind, str(device)) + extra + key
            obj = used.get(key) or self._upload_cache.get(key) or build()
            used[key] = obj
            return obj

        for inter_m, uf_m, if_m in batches:
            if uf_m.shape[1] != self.n_user_features or if_m.shape[1] != self.n_item_features:
                raise ValueError("feature matrices must keep the number of features the model was first fit with")
            if id(if_m) not in item_cache:
                item_cache[id(if_m)] = cached("features", if_m, (), lambda: SparseFeatures(if_m, device))
            uf = cached("features", uf_m, (), lambda: SparseFeatures(uf_m, device))
            itf = item_cache[id(if_m)]
            user_base = int(user_offset) + sum(b[1].shape[0] for b in dev_batches)
            inter = cached("interactions", inter_m, (uf.shape[0], itf.shape[0], user_base),
                           lambda: Interactions(inter_m, n_users=uf.shape[0], n_items=itf.shape[0], device=device))
            inter.user_base = user_base
            inter.dp_group = (self._dp_active(), self.process_group)
            dev_batches.append((inter, uf, itf))
        self._upload_cache = used          # only what this call used stays resident

        batched_alpha = calculate_batched_alpha(num_batches=len(dev_batches), alpha=alpha)
        if verbose:
            logging.info('Beginning fitting')

        # launch-bound steps (small batches: ~60 kernels of a few microseconds each) are captured in a HIP graph after one
        # eager execution and replayed; see _GraphedStep
        graphed = {}
        self._graph_pool_owner = []          # graphs captured by this call (they share the first one's memory pool)
        # (thread-local, previous mode restored on exit: two models fitting in different threads do not switch each other's
        # grouping, and a predict_top_k elsewhere keeps the counting sort -- ADVICE r2)
        dp = self._dp_active()
        if dp:
            # the exchange plan of this call is made at its first step, from the feature matrices the variables meet there
            self._dp_batches = dev_batches
            for var in self._store.variables.values():
                var._trec_feats = None
            # A previous call with dp_sync_every_call=False left rows with their owners.  They stay there while every rank's
            # feature matrices span the same columns as in that call: the owner's rows are current on the owner, which is all a
            # step reads of a rank-disjoint table.  Any rank whose ranges moved (it may now read rows another rank stepped) makes
            # ALL ranks sync first -- one 4-byte all-reduce decides, so that the ranks agree on the collective.
            from . import sharding
            ranges = tuple((b[1].col_range, b[2].col_range) for b in dev_batches)
            if getattr(self, "_dp_stale_rows", False):
                moved = int(ranges != getattr(self, "_dp_last_ranges", None))
                if sharding.all_reduce_scalar(moved, device, self.process_group) > 0:
                    self.dp_sync()
            self._dp_last_ranges = ranges
            self._dp_prev_plan, self._dp_plan = getattr(self, "_dp_plan", None), None
        try:
            with ops.deterministic_grouping(bool(getattr(self, "deterministic", False))):
                self._run_epochs(epochs, dev_batches, graphed, learning_rate, alpha, batched_alpha, n_sampled_items, verbose)
        finally:
            if dp:
                self._dp_batches = None
                if self._dp_plan is None:
                    # no optimiser step was made (epochs=0, no batches, an exception before step 1): the previous plan still
                    # describes who owns the stale rows / slots
                    self._dp_plan, self._dp_prev_plan = self._dp_prev_plan, None
        if dp and getattr(self, "dp_sync_every_call", True):
            self.dp_sync()

    def _run_epochs(self, epochs, dev_batches, graphed, learning_rate, alpha, batched_alpha, n_sampled_items, verbose):
        coops = {} if self.loss_graph_factory.is_sample_based else None
        for epoch in range(epochs):
            for batch, (inter, uf, itf) in enumerate(dev_batches):
                # models that fit on chip: the whole step as one cooperative kernel (from the second step on: the first creates the
                # variables and their Adam slots)
                coop = coops.get(batch) if coops is not None else None
                if coop is None and coops is not None and batch not in coops and (epoch >= 1 or self._opt_step > 0):
                    coop = coops[batch] = _CoopStep.plan(self, inter, uf, itf, n_sampled_items)
                if coop is not None and coop.ok:
                    out = coop.run(self, learning_rate, batched_alpha, verbose)
                    if out is not False:
                        self.last_step_form = "coop"
                        loss, serial_predictions, wr_loss = out
                        if verbose:
                            logging.info('EPOCH {} BATCH {} loss = {}, weight_reg_l2_loss = {}, mean_pred = {}'.format(
                                epoch, batch, float(loss.mean()), alpha * wr_loss, float(serial_predictions.mean())))
                        continue
                step = graphed.get(batch)
                if step is None and epoch >= 1 and epochs - epoch >= 2 and batch not in graphed and \
                        self._graph_eligible(inter, n_sampled_items, verbose):
                    step = graphed[batch] = _GraphedStep.capture(self, inter, uf, itf, n_sampled_items, learning_rate,
                                                                 batched_alpha)
                self.last_step_form = "graph" if step else "eager"
                if step:
                    loss, serial_predictions, wr_loss = step.run(self, verbose)
                else:
                    loss, serial_predictions, wr_loss = self._train_step(inter, uf, itf, learning_rate, batched_alpha,
                                                                         n_sampled_items, want_stats=verbose)
                if verbose:
                    mean_loss = float(loss.mean())
                    mean_pred = float(serial_predictions.mean())
                    weight_reg_l2_loss = alpha * wr_loss      # the reference logs the UNbatched alpha (:631)
                    logging.info('EPOCH {} BATCH {} loss = {}, weight_reg_l2_loss = {}, mean_pred = {}'.format(
                        epoch, batch, mean_loss, weight_reg_l2_loss, mean_pred
                    ))

    def _schedule_state(self):
        """float32[4] on the device: {beta1_power, beta2_power, lr_t, sample step bits} for the CURRENT host counters."""
        if getattr(self, '_schedule', None) is None:
            self._schedule = torch.zeros((4,), dtype=torch.float32, device=self._store.device)
            self._schedule_mirror = None
        self._sync_schedule_state()
        return self._schedule

    def _sync_schedule_state(self):
        """Eager steps advance only the host counters; before a replay the device state is brought back in line."""
        if self._schedule_mirror == (self._opt_step, self._sample_step):
            return
        b1p, b2p = np.float32(1.0), np.float32(1.0)
        for _ in range(int(self._opt_step)):
            b1p = np.float32(b1p * np.float32(ADAM_BETA1))
            b2p = np.float32(b2p * np.float32(ADAM_BETA2))
        host = np.zeros(4, np.float32)
        host[0], host[1] = b1p, b2p
        host.view(np.uint32)[3] = np.uint32(self._sample_step & 0xFFFFFFFF)
        self._schedule.copy_(torch.from_numpy(host))
        self._schedule_mirror = (self._opt_step, self._sample_step)

    def _graph_eligible(self, inter, n_sampled_items, verbose):
        if not self.hip_graphs or self._dp_active() or self._capture is not None or getattr(self, 'deterministic', False):
            return False
        # only steps made of the library's own launches are captured: the built-in losses on built-in graphs.  User-defined
        # torch graphs / losses may do anything (host syncs, allocations the capture cannot follow) -- capturing them would
        # fail, warn and fall back on every fit call.
        if type(self.loss_graph_factory) not in _GRAPH_CAPTURABLE_LOSSES or not self._is_engine_graph():
            return False
        if len(self._graph_pool_owner) >= MAX_GRAPHED_BATCHES:
            return False
        pairs = inter.nnz + inter.shape[0] * int(n_sampled_items or 0)
        dense = self.loss_graph_factory.is_dense
        return pairs <= 4_000_000 and (not dense or inter.shape[0] * inter.shape[1] <= 4_000_000)

    def _draw_samples(self, inter, n_sampled_items):
        """tf.py_func(sample_items) of tensorrec.py:298-302: one [n_users, n_sampled_items] int32 table per step."""
        n_users, n_items = inter.shape
        self._sample_step += 1
        samples = self.sampler.sample(n_items, n_users, int(n_sampled_items),
                                      self.loss_graph_factory.is_sampled_with_replacement, self._sample_step,
                                      self._store.device, getattr(inter, 'user_base', 0))
        return samples.to(torch.int32).contiguous()

    def _train_step(self, inter, user_feats, item_feats, learning_rate, alpha, n_sampled_items, want_stats=False,
                    samples=None, apply=True):
        """One optimiser step.  ``samples``: a pre-drawn sample table (graph replays feed a static buffer); ``apply=False``
        stops after the backward pass and returns (loss, serial predictions, regularised weights, loss length)."""
        loss_graph = self.loss_graph_factory
        graph = self.prediction_graph_factory
        n_users, n_items = inter.shape
        with variable_scope(self._store):
            user_reprs, attn_reprs, item_repr, user_bias, item_bias, weights = \
                self._representations(user_feats, item_feats)
            user_repr = user_reprs[0]
            multi = self._multi()
            x_user = PairIndex.make(inter.x_user, inter.x_user32, interactions=inter)
            x_item = PairIndex.make(inter.x_item, inter.x_item32, interactions=inter)

            engine = self._is_engine_graph()
            # built-in WMRB on dot / cosine scores: one fused pass per user (csrc/wmrb_fused.hip) instead of
            # serial scores -> loss -> autograd; only the exact built-in classes qualify (subclasses may override)
            # (rows in registers: dot scores, S + the longest row <= 256; otherwise -- S in the thousands, Euclidean scores --
            # the tiled form of the same step, csrc/wmrb_tiled.hip)
            fused = tiled = False
            if (engine and not multi and graph.engine_mode in (ops.MODE_DOT, ops.MODE_EUCLIDEAN)
                    and type(loss_graph) in (WMRBLossGraph, BalancedWMRBLossGraph)
                    and n_sampled_items is not None and item_repr.dim() == 2 and user_reprs[0].dim() == 2
                    and user_reprs[0].shape[1] == item_repr.shape[1]):
                fused = graph.engine_mode == ops.MODE_DOT and \
                    ops.wmrb_fused_supported(n_sampled_items, inter, int(item_repr.shape[1]))
                tiled = not fused and ops.wmrb_tiled_supported(n_sampled_items, inter, int(item_repr.shape[1]))
            u_ins, a_ins, i_in = user_reprs, attn_reprs, item_repr
            if engine and graph.engine_normalize:     # cosine: normalise once, share between all serial calls
                u_ins = [ops.l2_normalize_rows(u) for u in user_reprs]
                a_ins = [ops.l2_normalize_rows(a) for a in attn_reprs] if attn_reprs is not None else None
                i_in = ops.l2_normalize_rows(item_repr)
            u_in = u_ins[0]
            if fused or tiled:
                return self._fused_wmrb_step(inter, u_in, i_in, user_bias, item_bias, weights, learning_rate, alpha,
                                             n_sampled_items, want_stats, samples, apply,
                                             tiled_mode=graph.engine_mode if tiled else None)
            if multi:
                pred_serial = self._serial_multi(u_ins, a_ins, i_in, x_user, x_item, user_bias, item_bias)
            elif engine:
                pred_serial = self._serial_engine(u_in, i_in, x_user, x_item, user_bias, item_bias)
            else:
                pred_serial = self._serial(user_repr, item_repr, x_user, x_item, user_bias, item_bias)

            loss_kwargs = {
                'tf_prediction_serial': pred_serial,
                'tf_interactions_serial': inter.values,
                'tf_interactions': inter,
                'tf_n_users': n_users,
                'tf_n_items': n_items,
            }
            if loss_graph.is_dense:
                # the built-in dense losses on dot / cosine scores need two Gram matrices, not the [n_users, n_items] tensor
                # (ops.FactoredPrediction, csrc/loss_dense.hip): O((U + I) d^2), and 1M x 1M -- 4 TB of predictions -- runs at all
                factored = (engine and not multi and graph.engine_mode == ops.MODE_DOT
                            and type(loss_graph) in (RMSEDenseLossGraph, SeparationDenseLossGraph)
                            and ops.N.load().trec_get_tuning(b"dense_loss_factored", 1) != 0)
                if factored:
                    tf_prediction = ops.FactoredPrediction(u_in, i_in, user_bias if self.biased else None,
                                                           item_bias if self.biased else None, pred_serial)
                elif multi:
                    tf_prediction = self._dense_multi(user_reprs, attn_reprs, item_repr, user_bias, item_bias,
                                                      differentiable=True)
                else:
                    tf_prediction = self._dense_prediction(user_repr, item_repr, user_bias, item_bias,
                                                           differentiable=True)
                # TF evaluates the rankings node only when a loss uses it, and no built-in loss does: the U * I^2 counting
                # kernel runs for custom loss graphs only (50 ms of a 54 ms RMSEDense step at 20000 x 5000)
                builtin = factored or type(loss_graph).connect_loss_graph.__module__ == AbstractLossGraph.__module__
                loss_kwargs.update({'tf_prediction': tf_prediction,
                                    'tf_rankings': None if builtin else rank_predictions(tf_prediction)})
            if loss_graph.is_sample_based:
                if samples is None:
                    samples = self._draw_samples(inter, n_sampled_items)
                if engine:
                    xs_item = PairIndex.make(samples.reshape(-1), samples.reshape(-1), int(n_sampled_items))
                    if multi:
                        samp_serial = self._serial_multi(u_ins, a_ins, i_in, xs_item, xs_item, user_bias, item_bias,
                                                         sampled=True)
                    else:
                        samp_serial = self._serial_engine(u_in, i_in, xs_item, xs_item, user_bias, item_bias)
                elif multi:
                    xs_user64 = torch.arange(n_users, device=samples.device).repeat_interleave(int(n_sampled_items))
                    xs_item64 = samples.reshape(-1).to(torch.int64)
                    samp_serial = self._serial_multi(user_reprs, attn_reprs, item_repr, PairIndex.make(xs_user64),
                                                     PairIndex.make(xs_item64), user_bias, item_bias, sampled=True)
                else:
                    xs_user64 = torch.arange(n_users, device=samples.device).repeat_interleave(int(n_sampled_items))
                    xs_item64 = samples.reshape(-1).to(torch.int64)
                    samp_serial = self._serial(user_repr, item_repr, PairIndex.make(xs_user64),
                                               PairIndex.make(xs_item64), user_bias, item_bias)
                loss_kwargs.update({
                    'tf_sample_predictions': densify_sampled_item_predictions(samp_serial, n_sampled_items, n_users),
                    'tf_n_sampled_items': n_sampled_items,
                })

            # (user shards: a scalar loss is ONE number over the union batch -- its sums cross the ranks inside the loss op)
            with ops.scalar_loss_group(self.process_group, active=self._dp_active()):
                basic_loss = loss_graph.connect_loss_graph(**loss_kwargs)

        # tf_loss = tf_basic_loss + alpha * reg (broadcast), minimised as a sum (tensorrec.py:487-489)
        n_loss = int(basic_loss.numel())
        basic_loss.sum().backward()
        if not apply:
            return basic_loss, pred_serial, weights, n_loss
        return self._apply_gradients(basic_loss, pred_serial, weights, n_loss, learning_rate, alpha, want_stats)

    def _fused_wmrb_step(self, inter, u_in, i_in, user_bias, item_bias, weights, learning_rate, alpha, n_sampled_items,
                         want_stats, samples=None, apply=True, tiled_mode=None):
        """The WMRB step with the user side in one kernel: loss values and d(sum loss)/d(representations, biases) come
        from ops.wmrb_fused_step; autograd then only carries them through the representation graphs (K1 backward)."""
        loss_graph = self.loss_graph_factory
        if samples is None:
            samples = self._draw_samples(inter, n_sampled_items)
        ub = user_bias if self.biased else None
        ib = item_bias if self.biased else None
        if tiled_mode is not None:
            basic_loss, pred_serial, d_u, d_v, d_ub, d_ib = ops.wmrb_tiled_step(u_in, i_in, ub, ib, inter, samples,
                                                                                balanced=loss_graph.balanced, mode=tiled_mode)
        else:
            basic_loss, pred_serial, d_u, d_v, d_ub, d_ib = ops.wmrb_fused_step(u_in, i_in, ub, ib, inter, samples,
                                                                                balanced=loss_graph.balanced)
        tensors, grads = [u_in, i_in], [d_u, d_v]
        if self.biased:
            tensors += [user_bias, item_bias]
            grads += [d_ub, d_ib]
        ops.accumulate_grads(tensors, grads)                    # (weight-less graphs have no history: skipped inside)
        if not apply:
            return basic_loss, pred_serial, weights, int(basic_loss.numel())
        return self._apply_gradients(basic_loss, pred_serial, weights, int(basic_loss.numel()), learning_rate, alpha,
                                     want_stats)

    def _apply_gradients(self, basic_loss, pred_serial, weights, n_loss, learning_rate, alpha, want_stats,
                         static_grads=None):
        """Gradient all-reduce (data-parallel fit), the L2 term and the fused Adam step (tensorrec.py:487-489)."""
        loss_graph = self.loss_graph_factory

        if self._dp_active():
            return self._dp_apply_gradients(basic_loss, pred_serial, weights, n_loss, learning_rate, alpha, want_stats)

        reg_ids = set(id(w) for w in weights)
        if self._capture is not None:
            self._capture['loss'] = basic_loss.detach().cpu().numpy().copy()
            self._capture['pred_serial'] = pred_serial.detach().cpu().numpy().copy()
            self._capture['grads'] = {n: (v.grad.detach().cpu().numpy().copy() if v.grad is not None else None)
                                      for n, v in self._store.variables.items()}
        self._opt_step += 1
        lr_t = _adam_lr_t(learning_rate, self._opt_step)
        l2 = float(np.float32(np.float32(n_loss) * np.float32(alpha)))
        for name in self._store.order:
            var = self._store.variables[name]
            if name not in self._adam:
                self._adam[name] = (torch.zeros_like(var), torch.zeros_like(var))
            m, v = self._adam[name]
            grad = static_grads.get(name) if static_grads is not None else var.grad
            if grad is None:
                grad = torch.zeros_like(var)
            with torch.no_grad():
                ops.adam_tf_step(var, m, v, grad, lr_t, l2 if id(var) in reg_ids else 0.0, ADAM_BETA1, ADAM_BETA2,
                                 ADAM_EPSILON)
            var.grad = None

        if want_stats:
            with torch.no_grad():
                wr = float(sum(0.5 * float((w.detach() ** 2).sum()) for w in weights))
            return basic_loss.detach(), pred_serial.detach(), wr
        return None, None, None

    # ------------------------------------------------------------------------------------------ data-parallel step
    def _dp_make_plan(self):
        """The gradient-exchange plan of this fit call (sharding.plan_gradient_exchange; a COLLECTIVE, at the call's first
        step): a variable that met only this call's user (or item) feature matrices -- sparse_dense_matmul / sparse_matvec
        mark it -- can receive gradient only in the rows those matrices have columns for, over ALL of the call's batches."""
        from . import sharding
        store = self._store
        batches = getattr(self, "_dp_batches", None) or []
        supports = {}
        if batches:
            uf0, itf0 = id(batches[0][1]), id(batches[0][2])
            for name in store.order:
                met = getattr(store.variables[name], "_trec_feats", None)
                if not met or not met <= {uf0, itf0}:
                    continue
                ranges = [b[1].col_range for b in batches] if uf0 in met else []
                ranges += [b[2].col_range for b in batches] if itf0 in met else []
                ranges = [r for r in ranges if r[1] > r[0]]
                supports[name] = (min(r[0] for r in ranges), max(r[1] for r in ranges)) if ranges else (0, 0)
        shapes = {name: tuple(store.variables[name].shape) for name in store.order}
        plan = sharding.plan_gradient_exchange(store.order, shapes, supports, store.device, self.process_group)
        # the marks are structural, the check is not: a gradient outside the claimed rows (a table ALSO used some other way)
        # would be lost silently -- one look at the first step's gradients, once per call; the ranks agree on the outcome and
        # such a table is exchanged in full
        names = [n for n in store.order if plan.mode[n] == "disjoint"]
        if names:
            bad = torch.zeros((len(names),), dtype=torch.float32, device=store.device)
            for j, name in enumerate(names):
                g, (lo, hi) = store.variables[name].grad, supports[name]
                if g is not None:
                    bad[j] = float(bool(g[:lo].any().item()) or bool(g[hi:].any().item()))
            bad = sharding.all_reduce_max(bad, self.process_group).cpu().numpy()
            if bad.any():
                for j, name in enumerate(names):
                    if bad[j]:
                        supports.pop(name)
                plan = sharding.plan_gradient_exchange(store.order, shapes, supports, store.device, self.process_group)
        # Adam slots left with their owners under ANOTHER ownership: bring them together (under the old plan) first
        prev = getattr(self, "_dp_prev_plan", None)
        if prev is not None and prev.key != plan.key and (getattr(self, "_dp_stale_slots", False) or
                                                          getattr(self, "_dp_stale_rows", False)):
            self._dp_plan = prev
            self.dp_sync(optimizer_state=True)
        self._dp_prev_plan = None
        return plan

    def _dp_apply_gradients(self, basic_loss, pred_serial, weights, n_loss, learning_rate, alpha, want_stats):
        """One optimiser step of the user-sharded fit (sharding.py, "gradient exchange"): the objective is a sum over
        interactions, so the union batch's gradient is the sum of the shard gradients.  Rank-disjoint tables: no exchange, the
        owner steps its rows.  Large shared tables: reduce-scatter -> Adam on the owned rows -> all-gather, the exchange in
        flight (RCCL's stream) while the other tables are stepped.  Small tensors: all-reduce + the same Adam everywhere."""
        import torch.distributed as dist
        from . import sharding
        loss_graph = self.loss_graph_factory
        scalar = basic_loss.dim() == 0
        if scalar and type(loss_graph).connect_loss_graph.__module__ != AbstractLossGraph.__module__ and \
                not getattr(self, "_dp_scalar_warned", False):
            # the built-in scalar losses all-reduce their sums (ops.scalar_loss_group): one number over the union batch, the
            # reference's value for user_batch_size=None.  What a user-defined scalar means across shards only its author knows.
            logging.warning("data-parallel fit with the user-defined scalar loss %s: the objective is the SUM of the ranks' "
                            "values unless the loss reduces over ops.scalar_loss_group itself" % type(loss_graph).__name__)
            self._dp_scalar_warned = True
        store, group = self._store, self.process_group
        for name in store.order:
            var = store.variables[name]
            if var.grad is None:
                var.grad = torch.zeros_like(var)
        if self._dp_plan is None:
            self._dp_plan = self._dp_make_plan()
        plan = self._dp_plan
        rank = dist.get_rank(group)
        # tf_loss = basic_loss + alpha * reg is summed over the loss entries (tensorrec.py:487-489): every interaction's for a
        # vector loss, ONE for a scalar
        n_loss = 1 if scalar else sharding.all_reduce_scalar(n_loss, store.device, group)
        if self._capture is not None:                       # (tests: the LOCAL gradients, before any exchange)
            self._capture['loss'] = basic_loss.detach().cpu().numpy().copy()
            self._capture['pred_serial'] = pred_serial.detach().cpu().numpy().copy()
            self._capture['grads'] = {n: v.grad.detach().cpu().numpy().copy() for n, v in store.variables.items()}
        self._opt_step += 1
        lr_t = _adam_lr_t(learning_rate, self._opt_step)
        l2 = float(np.float32(np.float32(n_loss) * np.float32(alpha)))
        reg_ids = set(id(w) for w in weights)
        # ---- the exchanges leave first
        pending = {}
        for name in store.order:
            var = store.variables[name]
            if plan.mode[name] == "sharded":
                pending[name] = sharding.reduce_scatter_rows(var.grad, plan.bounds[name], rank, group, async_op=True)
            elif plan.mode[name] == "replicated" and sharding.active(group):
                pending[name] = (var.grad, dist.all_reduce(var.grad, op=dist.ReduceOp.SUM, group=group, async_op=True))

        def step(name, lo=None, hi=None, grad=None):
            var = store.variables[name]
            if name not in self._adam:
                self._adam[name] = (torch.zeros_like(var), torch.zeros_like(var))
            m, v = self._adam[name]
            w = var.detach()
            if lo is not None:
                w, m, v = w[lo:hi], m[lo:hi], v[lo:hi]
            if w.numel():
                ops.adam_tf_step(w, m, v, grad, lr_t, l2 if id(var) in reg_ids else 0.0, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON)

        # ---- tables that need nobody: the owner steps its rows
        with torch.no_grad():
            for name in store.order:
                if plan.mode[name] == "disjoint":
                    lo, hi = plan.own[name]
                    step(name, lo, hi, store.variables[name].grad[lo:hi])
                    self._dp_stale_rows = self._dp_stale_slots = True
            gathers = []
            for name in store.order:
                if name not in pending:
                    if plan.mode[name] == "replicated":
                        step(name, grad=store.variables[name].grad)
                    continue
                grad, work = pending[name]
                if work is not None:
                    work.wait()
                if plan.mode[name] == "replicated":
                    step(name, grad=grad)
                else:
                    lo, hi = plan.own[name]
                    step(name, lo, hi, grad)
                    gathers.append(sharding.all_gather_rows(store.variables[name].detach(), plan.bounds[name], rank, group,
                                                            async_op=True))
                    self._dp_stale_slots = True               # (the Adam slots of the rows this rank does not own)
            for work in gathers:
                if work is not None:
                    work.wait()
        for var in store.variables.values():
            var.grad = None
        if want_stats:
            self.dp_sync()
            with torch.no_grad():
                wr = float(sum(0.5 * float((w.detach() ** 2).sum()) for w in weights))
            return basic_loss.detach(), pred_serial.detach(), wr
        return None, None, None

    @_on_model_device
    def dp_sync(self, optimizer_state=False):
        """Data-parallel fit: bring this replica's copy of the rows other ranks own up to date -- the weights of rank-disjoint
        tables (one broadcast per owner); with ``optimizer_state`` also every Adam slot a rank stepped alone (rank-disjoint and
        sharded tables).  A COLLECTIVE over the model's process group: every rank must call it.  ``fit_partial`` calls it at its
        end unless ``dp_sync_every_call=False``; ``save_model`` needs ``dp_sync(optimizer_state=True)`` first."""
        from . import sharding
        plan = getattr(self, "_dp_plan", None)
        if not sharding.active(self.process_group):
            self._dp_stale_rows = self._dp_stale_slots = False
            return
        if plan is None:
            if self._dp_stale_rows or self._dp_stale_slots:
                raise RuntimeError("dp_sync: rows / Adam slots are with their owners but the exchange plan that says who owns "
                                   "them is gone (a model copied or unpickled between a fit call and its dp_sync?)")
            return
        store = self._store
        for name in store.order:
            mode = plan.mode.get(name)
            tensors = []
            if mode == "disjoint" and self._dp_stale_rows:
                tensors.append(store.variables[name].detach())
            if mode in ("disjoint", "sharded") and optimizer_state and self._dp_stale_slots and name in self._adam:
                tensors += list(self._adam[name])
            if tensors:
                sharding.sync_owned_rows(tensors, plan.bounds[name], self.process_group)
        self._dp_stale_rows = False
        if optimizer_state:
            self._dp_stale_slots = False

    # ------------------------------------------------------------------------------------------ predict
    def clear_predict_cache(self):
        """Drop the device copies of the feature matrices the predict* calls keep between calls (at most four matrices and
        PREDICT_CACHE_MAX_NNZ non-zeros, ~4 GB; kept across fit calls on purpose: the reference's idiom is
        ``for epoch: fit_partial(epochs=1); evaluate``)."""
        self.__dict__['_predict_cache'] = {}

    def _check_fit(self, method):
        if self._store is None:
            raise ModelNotFitException(method=method)

    def _inference(self, user_features=None, item_features=None):
        """Device copies of the feature matrices of a predict* call.  Matrices whose CONTENT the previous predict* calls
        uploaded are reused (the fingerprint of ``fit``'s upload cache: xxh3 of the CSR arrays, ~3 ms for 1M identity rows):
        serving the same catalogue call after call uploads it once."""
        device = self._store.device

        def one(raw):
            if raw is None:
                return None
            m = self._single(raw)
            key = _fingerprint(m) if self.cache_uploads else None
            if key is None:
                return SparseFeatures(m, device)
            cache = self.__dict__.setdefault('_predict_cache', {})
            key = (str(device),) + key
            obj = cache.pop(key, None)
            if obj is None:
                obj = SparseFeatures(m, device)
            cache[key] = obj                              # most recently used last
            # at most four matrices and PREDICT_CACHE_MAX_NNZ non-zeros stay resident (device CSR arrays + cached transposes:
            # ~20 B per non-zero); a matrix beyond the budget on its own is used for this call and not kept
            while len(cache) > 4 or (len(cache) > 1 and sum(f.nnz for f in cache.values()) > PREDICT_CACHE_MAX_NNZ):
                cache.pop(next(iter(cache)))
            if obj.nnz > PREDICT_CACHE_MAX_NNZ:
                cache.pop(key, None)
            return obj
        return one(user_features), one(item_features)

    def _single(self, raw):
        mats = self._as_list(raw)
        return mats[0] if len(mats) == 1 else sp.vstack(mats, format='csr')

    def _predict_device(self, user_features, item_features):
        uf, itf = self._inference(user_features, item_features)
        with torch.no_grad(), variable_scope(self._store):
            user_reprs, attn_reprs, item_repr, user_bias, item_bias, _ = self._representations(uf, itf)
            if self._multi():
                return self._dense_multi(user_reprs, attn_reprs, item_repr, user_bias, item_bias)
            return self._dense_prediction(user_reprs[0], item_repr, user_bias, item_bias)

    @_on_model_device
    def predict(self, user_features, item_features):
        """Recommendation scores, ndarray [n_users, n_items] float32 (tensorrec.py:636-664)."""
        self._check_fit('predict')
        return _to_host(self._predict_device(user_features, item_features))

    @_on_model_device
    def predict_rank(self, user_features, item_features):
        """Recommendation ranks, ndarray [n_users, n_items] int32, 1 = best (tensorrec.py:705-733)."""
        self._check_fit('predict_rank')
        pred = self._predict_device(user_features, item_features)
        return _to_host(rank_predictions(pred))

    @_on_model_device
    def predict_rank_of_interactions(self, user_features, item_features, interactions, user_batch_size=None):
        """EXTENSION: the ranks ``predict_rank`` would give, but only at the positive entries of ``interactions`` --
        all the evaluation metrics need (eval.py multiplies the [n_users, n_items] rank matrix by the positive mask).
        Users are walked in tiles: a [tile, n_items] score slab stays on the device and K4 ranks each positive pair
        against its row (``trec_rank_of_pairs``, or the row-sorting ``trec_rank_rows`` when users have many positives),
        so neither scores nor ranks of the full matrix ever reach the host.  Returns ``eval.PairRanks`` (accepted by every metric in place of the matrix);
        the ranks are bit-identical to ``predict_rank(...)[rows, cols]``."""
        from .eval import PairRanks
        self._check_fit('predict_rank_of_interactions')
        uf, itf = self._inference(user_features, item_features)
        m = sp.csr_matrix(interactions)
        m.sort_indices()
        if m.shape[0] > uf.shape[0] or m.shape[1] > itf.shape[0]:
            raise ValueError("interactions do not fit the feature matrices")
        # serial (row-major) view of the positive entries, int32 throughout; the all-positive case copies nothing
        rows = np.repeat(np.arange(m.shape[0], dtype=np.int32), np.diff(m.indptr))
        cols, vals = m.indices.astype(np.int32, copy=False), m.data
        pos = vals > 0
        if not pos.all():
            rows, cols, vals = rows[pos], cols[pos], vals[pos]
        n_users, n_items = uf.shape[0], itf.shape[0]
        if user_batch_size is None:
            user_batch_size = max(64, min(n_users, (1 << 29) // max(1, n_items)))       # <= 2 GB of fp32 scores
        device = self._store.device
        ranks = np.zeros(len(rows), np.int32)
        bounds = np.searchsorted(rows, np.arange(0, n_users + user_batch_size, user_batch_size))
        # one taste, built-in prediction graph, fp32: no slab at all -- the count is the epilogue of the fp32 MFMA score
        # kernel (csrc/score_rank.hip), the targets' exact scores come from the same fmaf chain
        fused = (self._is_engine_graph() and not self._multi() and self.precision == 'fp32' and len(rows) > 0 and
                 self.n_components <= 256 and N.load().trec_get_tuning(b"rank_fused", 1) != 0)
        with torch.no_grad(), variable_scope(self._store):
            user_reprs, attn_reprs, item_repr, user_bias, item_bias, _ = self._representations(uf, itf)
            if fused:
                graph = self.prediction_graph_factory
                want_sq = graph.engine_mode == ops.MODE_EUCLIDEAN
                u_op, u_sq, kpad = ops.score_prep(user_reprs[0], ops.DTYPE_F32, normalize=graph.engine_normalize,
                                                  want_sqnorm=want_sq)
                i_op, i_sq, _ = ops.score_prep(item_repr, ops.DTYPE_F32, normalize=graph.engine_normalize,
                                               want_sqnorm=want_sq)
                if want_sq and u_op.shape[1] != kpad:          # (score_prep returns the representation itself when d == kpad)
                    raise RuntimeError("score_prep returned an unpadded operand")
                pair_indptr = np.searchsorted(rows, np.arange(n_users + 1)).astype(np.int64)
                counts = ops.rank_counts_fused(u_op, i_op, kpad, user_reprs[0].shape[1], pair_indptr,
                                               torch.from_numpy(np.ascontiguousarray(cols)).to(device),
                                               user_bias.contiguous() if self.biased else None,
                                               item_bias.contiguous() if self.biased else None, graph.engine_mode,
                                               u_sq, i_sq)
                return PairRanks(rows, (counts + 1).cpu().numpy(), vals, n_users)
            for b, s in enumerate(range(0, n_users, user_batch_size)):
                p0, p1 = bounds[b], bounds[b + 1]
                if p0 == p1:
                    continue
                e = min(s + user_batch_size, n_users)
                ub = user_bias[s:e] if user_bias is not None else None
                if self._multi():
                    slab = self._dense_multi([u[s:e] for u in user_reprs],
                                             [a[s:e] for a in attn_reprs] if attn_reprs is not None else None,
                                             item_repr, ub, item_bias)
                else:
                    slab = self._dense_prediction(user_reprs[0][s:e], item_repr, ub, item_bias)
                slab = slab.contiguous()
                xi32 = torch.from_numpy(cols[p0:p1]).to(device)
                xu = (torch.from_numpy(rows[p0:p1]).to(device) - s).long()
                xi = xi32.long()
                if 64 <= n_items <= 32768 and (p1 - p0) >= 32 * (e - s):
                    # many positives per user: sort every row once (K4's sorted form, ~0.1 ms per 32768-item row and CU)
                    # and read the pairs' ranks off the tile, instead of one 26k-item count per pair
                    r = ops.rank_rows(slab)[xu, xi]
                else:
                    # (rows is sorted: the tile's pairs are grouped by user; one pass over a row serves all its targets)
                    target = slab[xu, xi].contiguous()
                    tile_ptr = torch.from_numpy(np.searchsorted(rows[p0:p1], np.arange(s, e + 1)).astype(np.int64)).to(device)
                    r = ops.rank_of_pairs_by_user(slab, 0, 0, n_items, tile_ptr, xi32, target, add_one=True)
                ranks[p0:p1] = r.cpu().numpy()
        return PairRanks(rows, ranks, vals, n_users)

    @_on_model_device
    def predict_top_k(self, user_features, item_features, k=10, user_batch_size=None, return_device=False,
                      item_sharded=False, item_offset=0, return_route=False):
        """EXTENSION: the k best items per user -- (scores [n_users, k] float32, item ids [n_users, k] int32),
        ordered like the first k ranks of ``predict_rank`` -- computed by the fused MFMA score + top-k kernel without
        materialising [n_users, n_items] (which is 4 TB at 1M x 1M).  ``user_batch_size`` None (default): as many users per
        pass as the free device memory holds (ops.topk_user_batch: ~22 KB of workspace per user at 1M items, so 1M users are
        ONE pass on a 288 GB device -- the per-pass costs of the cascade are paid once).

        ``item_sharded=True`` (torch.distributed initialised, one process per GPU): ``item_features`` holds THIS rank's
        rows of the item feature matrix, ``item_offset`` the global id of its first row, ``user_features`` is the same
        on every rank.  Every rank scores its shard; large catalogues first agree on one top-k floor per user (the k largest
        lower bounds of every shard), then the per-shard lists are merged -- both exchanges as user-partitioned all-to-alls
        over RCCL (every rank finalises 1/world of the users, the finished lists are then all-gathered so that every rank
        returns all of them), as plain all-gathers on backends without a device all-to-all (sharding.py).  Euclidean scores:
        every rank certifies its own shard's first k (no floor exchange), then the same merge.

        Which of the routes the call took is kept in ``self.last_route`` (a dict: "route" = cascade_int8 | bf16_filter |
        euclid_certified | wide_cascade | two_stage | direct | slab (attention models, k > 16 off the wide routes), "k", "sharded",
        "user_batch_size", "n_items") and, with
        ``return_route=True``, returned as a third value -- a silently slower route is the likeliest regression of this method
        (tests/test_gpu_routes.py pins the route of every BASELINE.json configuration)."""
        from . import sharding
        self._check_fit('predict_top_k')
        if int(k) < 1:
            raise ValueError("predict_top_k needs k >= 1 (got %r)" % (k,))
        if not self._is_engine_graph():
            raise ValueError("predict_top_k needs a built-in prediction graph")
        graph = self.prediction_graph_factory
        uf, itf = self._inference(user_features, item_features)
        dtype = ops.DTYPE_BF16 if self.precision == 'bf16' else ops.DTYPE_F32
        want_sq = graph.engine_mode == ops.MODE_EUCLIDEAN
        # attention models: the softmax-weighted sum over tastes (recommendation_graphs.py:98-107) does not decompose into
        # per-taste top-k lists, but it IS independent per (user, item): score slabs of a few thousand users through the
        # collapse kernel (K9), exact ranks pick the k best of every row -- and item shards merge like any other top-k
        slab_route = self.attention_graph_factory is not None or self.n_components > ops.SCORE_KMAX
        import torch.distributed as dist
        sharded = bool(item_sharded) and sharding.active(self.process_group)
        method, floor_exchange = "auto", None
        if sharded:
            # every rank must take the same code path (the floor exchange is a collective): decide on the smallest shard
            smallest = torch.tensor([itf.shape[0]], dtype=torch.int64, device=self._store.device)
            dist.all_reduce(smallest, op=dist.ReduceOp.MIN, group=self.process_group)
            if int(smallest.item()) >= ops.TWO_STAGE_MIN_ITEMS:
                method = "two_stage"
                # RCCL: user-partitioned all-to-all (each rank receives 1/world of an all-gather's bytes); else all-gather
                a2a = sharding.a2a_available(smallest, self.process_group)
                floor_fn = sharding.shared_topk_floor_a2a if a2a else sharding.shared_topk_floor
                floor_exchange = lambda sel_max: floor_fn(sel_max, self.process_group)  # noqa: E731
            else:
                method = "direct"
        # precision='fp32' on a large catalogue: the same exact fp32 result, with the contraction done once on bf16 MFMA
        # as an error-bounded filter and only the survivors re-scored in fp32 (ops.score_topk_filtered)
        n_items_min = int(smallest.item()) if sharded else itf.shape[0]
        filtered = (dtype == ops.DTYPE_F32 and graph.engine_mode == ops.MODE_DOT and 1 <= k <= 16 and
                    n_items_min >= ops.TWO_STAGE_MIN_ITEMS and self.n_components <= 256 and
                    ops.N.load().trec_get_tuning(b"topk_bf16_filter", 1) != 0)
        # Euclidean scores (one taste, fp32): the same cascade finds the 16 NEAREST items of every user -- nearest = largest
        # u.i - r_i / 2 -- the reference's chain re-scores them and a per-user certificate decides (ops.score_topk_euclid_filtered)
        # Item shards: every rank certifies ITS shard's first k on its own (the certificate is local: "no other item of this
        # shard can enter these k places"), the exact per-shard lists merge like any others -- no shared floor, no collective
        # inside the route, so the ranks need not agree on who falls back.  Several tastes: the same per taste, then the merge
        # of the taste lists (max over tastes commutes with the monotone bias additions).
        # 13 <= k <= 48: the same with the 32 / 64 nearest items from the WIDE cascade's lists, where the int8 cascade runs.
        euclid_wide = (ops.EUCLID_CANDIDATES - 4 < k <= ops.EUCLID_WIDE_K_MAX and
                       ops.cascade_prefilter_for(self.n_components, n_items_min) == "int8" and ops.i8_user_classes_enabled())
        euclid_filtered = (dtype == ops.DTYPE_F32 and graph.engine_mode == ops.MODE_EUCLIDEAN and
                           (1 <= k <= ops.EUCLID_CANDIDATES - 4 or euclid_wide) and n_items_min >= ops.TWO_STAGE_MIN_ITEMS and
                           self.n_components <= 256 and
                           ops.N.load().trec_get_tuning(b"topk_euclid_filter", 1) != 0)
        # 17 <= k <= 64 on a catalogue the cascade runs on: the same int8 -> bf16 stages, 1,024 candidate slots per user and a
        # wave-per-user finish over every listed item (ops.score_topk_filtered_wide).  Item shards: every rank finds ITS shard's
        # exact first k on its own (local thresholds, no collective inside the route -- the ranks agree on taking it because the
        # smallest shard decides), the per-shard lists merge like any others.
        wide = (dtype == ops.DTYPE_F32 and graph.engine_mode == ops.MODE_DOT and 16 < k <= ops.WIDE_K_MAX and
                ops.cascade_prefilter_for(self.n_components, n_items_min) == "int8" and
                ops.N.load().trec_get_tuning(b"topk_bf16_filter", 1) != 0 and ops.i8_user_classes_enabled())
        # k beyond the 16 entries of the fused lists and off the wide routes (small catalogues, bf16 scores, k > 64 / 48 Euclidean):
        # exact fp32 score slabs and the k best of every row (ops.topk_from_scores) -- any k, places beyond the catalogue -inf / -1
        if int(k) > 16 and not wide and not euclid_filtered:
            slab_route = True
        stats_exchange = (lambda g: sharding.all_reduce_max(g, self.process_group)) if sharded else None
        # ... and on a catalogue of >= 262,144 items an int8 MFMA pass (exact integer arithmetic, proven bound) first decides
        # which (superblock, user) pairs the bf16 stage has to look at at all (csrc/topk_cascade.hip)
        prefilter = ops.cascade_prefilter_for(self.n_components, n_items_min * (dist.get_world_size(self.process_group) if sharded else 1)) \
            if filtered else None
        if user_batch_size is None:
            route = "wide" if (wide or (euclid_filtered and k > ops.EUCLID_CANDIDATES - 4)) else \
                ("cascade" if (filtered or euclid_filtered) else "two_stage")
            user_batch_size = ops.topk_user_batch(uf.shape[0], itf.shape[0], self.n_components, self._store.device,
                                                  route=route, k=k)
            if sharded:                  # every rank walks the SAME user batches (each batch holds collectives): the smallest wins
                ubs = torch.tensor([user_batch_size], dtype=torch.int64, device=self._store.device)
                dist.all_reduce(ubs, op=dist.ReduceOp.MIN, group=self.process_group)
                user_batch_size = int(ubs.item())
        if slab_route:
            route_name = "slab"
        elif euclid_filtered:
            route_name = "euclid_certified"
        elif filtered:
            route_name = "cascade_int8" if prefilter == "int8" else "bf16_filter"
        elif wide:
            route_name = "wide_cascade"
        else:
            route_name = method if method != "auto" else ("two_stage" if itf.shape[0] >= ops.TWO_STAGE_MIN_ITEMS else "direct")
        self.last_route = {"route": route_name, "k": int(k), "sharded": bool(sharded), "n_items": int(itf.shape[0]),
                           "user_batch_size": int(user_batch_size), "precision": self.precision}
        _ret = (lambda v_, i_: (v_, i_, dict(self.last_route))) if return_route else (lambda v_, i_: (v_, i_))
        vals, idx = [], []
        if slab_route:
            # (also: representations wider than the fused kernels' resident operand -- K-looped fp32 GEMM slabs)
            planes = 1 + (2 * self.n_tastes if self.attention_graph_factory is not None else
                          (self.n_tastes if self._multi() else 0))
            step = max(1, min(int(user_batch_size), (1 << 28) // max(1, itf.shape[0] * planes)))
            if sharded:                  # (each batch ends in a collective: every rank takes the same steps)
                st = torch.tensor([step], dtype=torch.int64, device=self._store.device)
                dist.all_reduce(st, op=dist.ReduceOp.MIN, group=self.process_group)
                step = int(st.item())
            with torch.no_grad(), variable_scope(self._store):
                user_reprs, attn_reprs, item_repr, user_bias, item_bias, _ = self._representations(uf, itf)
                for s in range(0, uf.shape[0], step):
                    e = min(s + step, uf.shape[0])
                    ub = user_bias[s:e] if user_bias is not None else None
                    if self._multi():
                        attn = [a[s:e] for a in attn_reprs] if attn_reprs is not None else None
                        slab = self._dense_multi([u[s:e] for u in user_reprs], attn, item_repr, ub, item_bias)
                    else:
                        slab = self._dense_prediction(user_reprs[0][s:e], item_repr, ub, item_bias)
                    v, i = ops.topk_from_scores(slab.contiguous(), k)
                    i = torch.where(i >= 0, i + int(item_offset), i)
                    if sharded:
                        if sharding.a2a_available(v, self.process_group):
                            v, i = sharding.sharded_top_k_a2a(v, i, k, self.process_group, replicate=True)
                        else:
                            v, i = sharding.sharded_top_k(v, i, k, self.process_group)
                    vals.append(v)
                    idx.append(i)
            vals, idx = torch.cat(vals), torch.cat(idx)
            return _ret(vals, idx) if return_device else _ret(_to_host(vals), _to_host(idx))
        with torch.no_grad(), variable_scope(self._store):
            user_reprs, _, item_repr, user_bias, item_bias, _ = self._representations(uf, itf)
            ib = item_bias.contiguous() if self.biased else None
            if filtered or wide:
                i_f = ops.score_prep_filter(item_repr, normalize=graph.engine_normalize, bias=ib, want_gstats=True)
            if not filtered and not wide:         # (the wide route works on the filter operand alone)
                i_op, i_sq, kpad = ops.score_prep(item_repr, dtype, normalize=graph.engine_normalize, want_sqnorm=want_sq)
            s = 0
            while s < uf.shape[0]:
                e = min(s + user_batch_size, uf.shape[0])
                retry = False
                try:
                    v, i = self._topk_user_batch(s, e, user_reprs, item_repr, user_bias, ib, k, graph, dtype, want_sq, filtered,
                                                 euclid_filtered, prefilter, sharded, method, floor_exchange, stats_exchange,
                                                 item_offset, i_f if (filtered or wide) else None,
                                                 None if (filtered or wide) else (i_op, i_sq, kpad), wide=wide)
                except torch.cuda.OutOfMemoryError:
                    # the workspace model of ops.topk_user_batch was too optimistic for this device's state: half the users per
                    # pass (item shards: the ranks walk the same batches and a rank cannot shrink alone -- the error stands)
                    if sharded or e - s <= 4096:
                        raise
                    retry = True
                if retry:
                    # (outside the except block: the traceback no longer pins the failed call's tensors, so the cache really is
                    # returned; halved from the size that RAN -- the nominal one may exceed the users that were left)
                    torch.cuda.empty_cache()
                    user_batch_size = max(4096, min(user_batch_size, e - s) // 2)
                    continue
                vals.append(v)
                idx.append(i)
                s = e
        vals, idx = torch.cat(vals), torch.cat(idx)
        self.last_route["user_batch_size"] = int(user_batch_size)          # (after any out-of-memory halving)
        if return_device:
            return _ret(vals, idx)
        return _ret(_to_host(vals), _to_host(idx))

    def _topk_user_batch(self, s, e, user_reprs, item_repr, user_bias, ib, k, graph, dtype, want_sq, filtered, euclid_filtered,
                         prefilter, sharded, method, floor_exchange, stats_exchange, item_offset, i_f, i_ops, wide=False):
        """Users [s, e) of predict_top_k: every taste's exact top-k, merged, and (item shards) exchanged."""
        from . import sharding
        import torch.distributed as dist
        ub = user_bias[s:e].contiguous() if self.biased else None
        if i_ops is not None:
            i_op, i_sq, kpad = i_ops
        per_taste = []
        for user_repr in user_reprs:
            if euclid_filtered:
                per_taste.append(ops.score_topk_euclid_filtered(user_repr[s:e], item_repr, k, ub, ib,
                                                                item_index_base=int(item_offset)))
                continue
            if wide:
                u_f = ops.score_prep_filter(user_repr[s:e], normalize=graph.engine_normalize, sort_users=True, k=k, user_bias=ub)
                per_taste.append(ops.score_topk_filtered_wide(u_f, i_f, k, ub, ib, item_index_base=int(item_offset)))
                continue
            if filtered:
                u_f = ops.score_prep_filter(user_repr[s:e], normalize=graph.engine_normalize,
                                            sort_users=prefilter == "int8", k=k, user_bias=ub)
                # (item shards of >= 4 ranks: a user lists ~27 / world candidates per shard -> four users per wave)
                lanes = 16 if sharded and dist.get_world_size(self.process_group) >= 4 else 0
                per_taste.append(ops.score_topk_filtered(u_f, i_f, k, ub, ib, item_index_base=int(item_offset),
                                                         floor_exchange=floor_exchange,
                                                         stats_exchange=stats_exchange, prefilter=prefilter,
                                                         finish_lanes=lanes))
                continue
            u_op, u_sq, _ = ops.score_prep(user_repr[s:e], dtype, normalize=graph.engine_normalize,
                                           want_sqnorm=want_sq)
            per_taste.append(ops.score_topk(u_op, i_op, dtype, kpad, k, ub, ib, graph.engine_mode, u_sq, i_sq,
                                            item_index_base=int(item_offset), method=method,
                                            floor_exchange=floor_exchange))
        v, i = per_taste[0] if len(per_taste) == 1 else _merge_taste_topk(per_taste, k)
        if sharded:
            if sharding.a2a_available(v, self.process_group):
                v, i = sharding.sharded_top_k_a2a(v, i, k, self.process_group, replicate=True)
            else:
                v, i = sharding.sharded_top_k(v, i, k, self.process_group)
        return v, i

    @_on_model_device
    def predict_similar_items(self, item_features, item_ids, n_similar):
        """Most similar items, list of lists of (item_id, score) (tensorrec.py:666-703); the query item itself is
        included, as in the reference."""
        self._check_fit('predict_similar_items')
        _, itf = self._inference(None, item_features)
        with torch.no_grad(), variable_scope(self._store):
            item_repr, _ = self.item_repr_graph_factory.connect_representation_graph(
                tf_features=itf, n_components=self.n_components, n_features=self.n_item_features,
                node_name_ending='item')
            sims = predict_similar_items(self.prediction_graph_factory, item_repr, np.array(item_ids)).cpu().numpy()
        results = []
        for i in range(len(item_ids)):
            item_sims = sims[i]
            best = np.argpartition(item_sims, -n_similar)[-n_similar:]
            item_results = sorted(zip(best, item_sims[best]), key=lambda x: -x[1])
            results.append(item_results)
        return results

    @_on_model_device
    def predict_user_representation(self, user_features):
        """ndarray [n_users, n_components], or [n_tastes, n_users, n_components] when n_tastes > 1
        (tensorrec.py:735-762)."""
        self._check_fit('predict_user_representation')
        uf, _ = self._inference(user_features, None)
        with torch.no_grad(), variable_scope(self._store):
            user_reprs, _, _ = self._user_representations(uf)
        user_repr = _to_host(torch.stack(user_reprs))
        return user_repr[0] if self.n_tastes == 1 else user_repr

    @_on_model_device
    def predict_user_attention_representation(self, user_features):
        """ndarray [n_tastes, n_users, n_components] (tensorrec.py:764-793)."""
        self._check_fit('predict_user_attention_representation')
        if self.attention_graph_factory is None:
            raise ModelWithoutAttentionException()
        uf, _ = self._inference(user_features, None)
        with torch.no_grad(), variable_scope(self._store):
            _, attn_reprs, _ = self._user_representations(uf)
        return _to_host(torch.stack(attn_reprs))

    @_on_model_device
    def predict_item_representation(self, item_features):
        """ndarray [n_items, n_components] (tensorrec.py:795-816)."""
        self._check_fit('predict_item_representation')
        _, itf = self._inference(None, item_features)
        with torch.no_grad(), variable_scope(self._store):
            item_repr, _ = self.item_repr_graph_factory.connect_representation_graph(
                tf_features=itf, n_components=self.n_components, n_features=self.n_item_features,
                node_name_ending='item')
        return _to_host(item_repr)

    @_on_model_device
    def predict_user_bias(self, user_features):
        """ndarray [n_users] (tensorrec.py:818-842)."""
        self._check_fit('predict_user_bias')
        if not self.biased:
            raise ModelNotBiasedException(actor='user')
        uf, _ = self._inference(user_features, None)
        with torch.no_grad(), variable_scope(self._store):
            _, proj = project_biases(uf, self.n_user_features, name='user_feature_biases')
        return proj.cpu().numpy()

    @_on_model_device
    def predict_item_bias(self, item_features):
        """ndarray [n_items] (tensorrec.py:844-868)."""
        self._check_fit('predict_item_bias')
        if not self.biased:
            raise ModelNotBiasedException(actor='item')
        _, itf = self._inference(None, item_features)
        with torch.no_grad(), variable_scope(self._store):
            _, proj = project_biases(itf, self.n_item_features, name='item_feature_biases')
        return proj.cpu().numpy()

    # ------------------------------------------------------------------------------------------ weights access
    @_on_model_device
    def get_weights(self):
        """name -> ndarray copy of every variable (EXTENSION; used by the parity tests to share weights with the
        oracle, since the reference exposes no seed)."""
        self._check_fit('get_weights')
        return {k: v.detach().cpu().numpy().copy() for k, v in self._store.variables.items()}

    @_on_model_device
    def set_weights(self, weights, reset_optimizer=True):
        self._check_fit('set_weights')
        for k, arr in weights.items():
            var = self._store.variables[k]
            with torch.no_grad():
                var.copy_(torch.as_tensor(np.asarray(arr, np.float32).reshape(tuple(var.shape)), device=var.device))
        if reset_optimizer:
            self._adam = {}
            self._opt_step = 0
            self._dp_stale_slots = False            # (there are no slots left to bring together)
        # every rank sets the same weights (the caller's contract under data_parallel): nothing waits for an owner any more
        if set(weights.keys()) >= set(self._store.order):
            self._dp_stale_rows = False

    # ------------------------------------------------------------------------------------------ persistence
    def __getstate__(self):
        """The python object without device state (the role of ``_break_graph_hooks``, tensorrec.py:247-257): weights
        and optimiser slots travel in the checkpoint file next to the pickle."""
        if getattr(self, "_dp_stale_rows", False) or getattr(self, "_dp_stale_slots", False):
            # (pickle / deepcopy bypass save_model's check: a copy made now would hold rows only their owners have current)
            raise RuntimeError("data-parallel fit left rows / Adam slots with their owners: call dp_sync(optimizer_state=True) "
                               "on every rank before pickling or copying the model")
        state = dict(self.__dict__)
        state['_store'] = None
        state['_adam'] = {}
        state['_capture'] = None
        state['_graph_pool_owner'] = []
        state['_schedule'] = None
        state['_schedule_mirror'] = None
        state['_upload_cache'] = {}
        state['_predict_cache'] = {}
        state['_dp_plan'] = None
        state['_dp_prev_plan'] = None
        state['_dp_batches'] = None
        state['_dp_last_ranges'] = None
        state['process_group'] = None
        return state

    @_on_model_device
    def save_model(self, directory_path):
        """Saves the model to files in the given directory (tensorrec.py:869-893): ``tensorrec.pkl`` (the python
        object: hyper-parameters and graph objects) and ``tensorrec_session.npz`` (every variable, its Adam slots and
        the step counters -- what the TF checkpoint holds in the reference)."""
        self._check_fit('save_model')
        if getattr(self, "_dp_stale_rows", False) or getattr(self, "_dp_stale_slots", False):
            raise RuntimeError("data-parallel fit left rows / Adam slots with the ranks that own them: call "
                               "model.dp_sync(optimizer_state=True) on EVERY rank before save_model")
        if not os.path.exists(directory_path):
            os.makedirs(directory_path)
        arrays = {}
        for i, name in enumerate(self._store.order):
            arrays['var/%d' % i] = self._store.variables[name].detach().cpu().numpy()
            if name in self._adam:
                arrays['adam_m/%d' % i] = self._adam[name][0].cpu().numpy()
                arrays['adam_v/%d' % i] = self._adam[name][1].cpu().numpy()
        arrays['names'] = np.array(self._store.order)
        arrays['counters'] = np.array([self._opt_step, self._sample_step], dtype=np.int64)
        with open(os.path.join(directory_path, 'tensorrec_session.npz'), 'wb') as file:
            np.savez(file, **arrays)
        with open(os.path.join(directory_path, 'tensorrec.pkl'), 'wb') as file:
            pickle.dump(file=file, obj=self)

    @classmethod
    def load_model(cls, directory_path):
        """Loads the model saved in the given directory (tensorrec.py:895-917) onto this process's GPU; predictions and
        ranks are bit-identical to the saved model's and ``fit_partial`` continues from the saved optimiser state."""
        with open(os.path.join(directory_path, 'tensorrec.pkl'), 'rb') as file:
            model = pickle.load(file=file)
        device = model._device()
        with np.load(os.path.join(directory_path, 'tensorrec_session.npz'), allow_pickle=False) as ckpt, \
                torch.cuda.device(device):
            names = [str(n) for n in ckpt['names']]
            model._store = VariableStore(device, seed=model.seed)
            for i, name in enumerate(names):
                model._store.get(name, ckpt['var/%d' % i])
                if 'adam_m/%d' % i in ckpt.files:
                    model._adam[name] = (torch.from_numpy(ckpt['adam_m/%d' % i]).to(device),
                                         torch.from_numpy(ckpt['adam_v/%d' % i]).to(device))
            model._opt_step, model._sample_step = (int(c) for c in ckpt['counters'])
        return model

    @_on_model_device
    def build(self, n_user_features, n_item_features):
        """EXTENSION: create the variables without a training step (weights then come from set_weights or the
        initialisers); fit_partial does this implicitly on its first call."""
        device = self._device()
        if self._store is None:
            self.n_user_features, self.n_item_features = int(n_user_features), int(n_item_features)
            self._store = VariableStore(device, seed=self.seed)
            if self.sampler is None:
                self.sampler = DeviceSampler(self.seed if self.seed is not None else 0)
            uf = SparseFeatures(sp.csr_matrix((1, n_user_features), dtype=np.float32), device)
            itf = SparseFeatures(sp.csr_matrix((1, n_item_features), dtype=np.float32), device)
            with torch.no_grad(), variable_scope(self._store):
                self._representations(uf, itf)
        return self


# (every built-in loss is made of this library's launches only: the separation losses lost their boolean-mask host syncs when
# they became streaming reductions, csrc/loss_dense.hip)
_GRAPH_CAPTURABLE_LOSSES = (RMSELossGraph, RMSEDenseLossGraph, WMRBLossGraph, BalancedWMRBLossGraph, SeparationLossGraph,
                            SeparationDenseLossGraph)
MAX_GRAPHED_BATCHES = 64        # graphs kept alive by one fit call; further batches run eagerly


class _GraphedStep(object):
    """One training step of one user batch as ONE HIP graph (torch.cuda.CUDAGraph over the launches this library issues
    on torch's current stream): schedule advance -> negative sampling -> forward -> loss -> every backward kernel through
    torch autograd -> the fused Adam step of every variable.  What changes from step to step lives in device memory:
    the model's schedule state {beta1_power, beta2_power, lr_t, sample step} (trec_adam_schedule_advance is the graph's
    first node; the sampler and Adam launches read it).  A sampler other than the DeviceSampler stays outside and fills
    a static table before each replay.  Capture happens after one eager execution of the same step, so every lazily
    built structure (transposed CSR, balanced weights, kernel attributes, Adam slots) already exists."""

    @classmethod
    def capture(cls, model, inter, uf, itf, n_sampled_items, learning_rate, alpha):
        self = cls()
        self.inter, self.S = inter, n_sampled_items
        store = model._store
        try:
            loss_graph = model.loss_graph_factory
            sample_based = loss_graph.is_sample_based
            self.device_sampler = sample_based and type(model.sampler) is DeviceSampler
            # a static table for samplers that run outside the graph (capture only records launches: content irrelevant)
            self.samples = torch.zeros((inter.shape[0], int(n_sampled_items)), dtype=torch.int32,
                                       device=store.device) if sample_based and not self.device_sampler else None
            state = model._schedule_state()
            for var in store.variables.values():
                var.grad = None
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # every graph of one fit call shares ONE memory pool (the batches run one after the other, so their
            # temporaries can alias) instead of pinning a private copy of every forward/backward temporary per batch
            owner = model._graph_pool_owner
            pool = owner[0].pool() if owner else None
            with torch.cuda.graph(self.graph, pool=pool):
                ops.adam_schedule_advance(state, learning_rate, ADAM_BETA1, ADAM_BETA2, bump_sample_step=sample_based)
                samples = self.samples
                if self.device_sampler:
                    samples = ops.sample_items_dev(inter.shape[0], inter.shape[1], int(n_sampled_items),
                                                   loss_graph.is_sampled_with_replacement, model.sampler.seed, state,
                                                   store.device, getattr(inter, 'user_base', 0))
                out = model._train_step(inter, uf, itf, 0.0, 0.0, n_sampled_items, samples=samples, apply=False)
                self.loss, self.pred_serial, self.weights, self.n_loss = out
                l2 = float(np.float32(np.float32(self.n_loss) * np.float32(alpha)))
                reg_ids = set(id(w) for w in self.weights)
                self.grads = {}
                for name in store.order:
                    var = store.variables[name]
                    m, v = model._adam[name]
                    grad = var.grad if var.grad is not None else torch.zeros_like(var)
                    self.grads[name] = grad
                    with torch.no_grad():
                        ops.adam_tf_step_dev(var, m, v, grad, state, l2 if id(var) in reg_ids else 0.0, ADAM_BETA1,
                                             ADAM_BETA2, ADAM_EPSILON)
            for var in store.variables.values():
                var.grad = None
            self.sample_based = sample_based
            owner.append(self.graph)
            return self
        except Exception as exc:      # capture is an optimisation: fall back to eager steps for this batch
            logging.warning('HIP graph capture of the training step failed (%r); running eagerly', exc)
            for var in store.variables.values():
                var.grad = None
            return False

    def run(self, model, want_stats):
        model._sync_schedule_state()
        if self.samples is not None:
            model._sample_step += 1               # the host-side sampler sees the same step number as an eager step
            table = model.sampler.sample(self.inter.shape[1], self.inter.shape[0], int(self.S),
                                         model.loss_graph_factory.is_sampled_with_replacement, model._sample_step,
                                         model._store.device, getattr(self.inter, 'user_base', 0))
            self.samples.copy_(table.to(torch.int32))
        elif self.sample_based:
            model._sample_step += 1
        self.graph.replay()
        model._opt_step += 1
        model._schedule_mirror = (model._opt_step, model._sample_step)
        if want_stats:
            with torch.no_grad():
                wr = float(sum(0.5 * float((w.detach() ** 2).sum()) for w in self.weights))
            return self.loss.detach(), self.pred_serial.detach(), wr
        return None, None, None


class _CoopStep(object):
    """One training step of one user batch as ONE cooperative kernel (csrc/step_coop.hip; trec_fit_step_coop): models that fit on
    chip -- BASELINE.json configs[1], 943 x 1,682, d = 64 -- are bound by launches, not by memory (~25 graph-replayed launches = 0.68 ms
    per epoch).  Covered: LinearRepresentation on identity user features, LinearRepresentation on any item features, DotProduct,
    WMRB / BalancedWMRB, one taste, single process.  Everything else keeps the multi-launch / HIP-graph step."""

    @classmethod
    def plan(cls, model, inter, uf, itf, n_sampled_items):
        from .representation_graphs import LinearRepresentationGraph
        N = ops.N
        if N.load().trec_get_tuning(b"fit_step_coop", 1) == 0 or model._dp_active() or model._capture is not None or \
                getattr(model, 'deterministic', False) or model._multi() or model.attention_graph_factory is not None:
            return None
        if type(model.user_repr_graph_factory) is not LinearRepresentationGraph or \
                type(model.item_repr_graph_factory) is not LinearRepresentationGraph or \
                type(model.prediction_graph_factory) is not DotProductPredictionGraph or \
                type(model.loss_graph_factory) not in (WMRBLossGraph, BalancedWMRBLossGraph) or not n_sampled_items:
            return None
        if not getattr(uf, "is_identity", False) or uf.shape[0] != inter.shape[0] or int(n_sampled_items) > inter.shape[1]:
            return None
        store = model._store
        names = ["linear_weights_user_0", "linear_weights_item"] + (["user_feature_biases", "item_feature_biases"] if model.biased else [])
        if set(store.variables) != set(names) or any(n not in model._adam for n in names):
            return None                                         # (the first eager step creates variables and Adam slots)
        n_users, n_items = inter.shape
        d = int(model.n_components)
        if store.variables[names[0]].shape != (n_users, d) or store.variables[names[1]].shape != (itf.shape[1], d):
            return None
        need = int(N.query("trec_fit_step_coop_workspace_floats", n_users, n_items, d, int(n_sampled_items), int(inter.max_row_nnz)))
        if need < 0:
            return None
        self = cls()
        self.inter, self.itf, self.S, self.names = inter, itf, int(n_sampled_items), names
        dev = store.device
        self.ws = torch.empty((need,), dtype=torch.float32, device=dev)
        self.loss = torch.empty((inter.n_positive,), dtype=torch.float32, device=dev)
        self.pred = torch.empty((inter.nnz,), dtype=torch.float32, device=dev)
        self.weight = inter.balanced_weight() if model.loss_graph_factory.balanced else None
        self.device_sampler = type(model.sampler) is DeviceSampler
        self.powers = None                                      # (opt step, beta1^t, beta2^t) as TF keeps them: float32 running products
        self.ok = True
        return self

    def _lr_t(self, lr, t):
        if self.powers is None or self.powers[0] != t - 1:
            b1p, b2p = np.float32(1.0), np.float32(1.0)
            for _ in range(int(t) - 1):
                b1p = np.float32(b1p * np.float32(ADAM_BETA1))
                b2p = np.float32(b2p * np.float32(ADAM_BETA2))
        else:
            _, b1p, b2p = self.powers
        b1p = np.float32(b1p * np.float32(ADAM_BETA1))
        b2p = np.float32(b2p * np.float32(ADAM_BETA2))
        self.powers = (t, b1p, b2p)
        return float(np.float32(np.float32(lr) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p)))

    def run(self, model, learning_rate, alpha, want_stats):
        """Returns (loss, serial predictions, weight-reg loss) like _train_step, or False when the device refused the launch (the
        caller falls back for good)."""
        N = ops.N
        store, inter, itf = model._store, self.inter, self.itf
        v = store.variables
        wu, wi = v[self.names[0]], v[self.names[1]]
        mu, vu = model._adam[self.names[0]]
        mi, vi = model._adam[self.names[1]]
        if model.biased:
            bu, bi = v[self.names[2]], v[self.names[3]]
            (bum, buv), (bim, biv) = model._adam[self.names[2]], model._adam[self.names[3]]
        else:
            bu = bi = bum = buv = bim = biv = None
        model._sample_step += 1
        samples = None
        if not self.device_sampler:
            samples = model.sampler.sample(inter.shape[1], inter.shape[0], self.S, False, model._sample_step, store.device,
                                           getattr(inter, 'user_base', 0)).to(torch.int32).contiguous()
        t = model._opt_step + 1
        lr_t = self._lr_t(learning_rate, t)
        l2 = float(np.float32(np.float32(inter.n_positive) * np.float32(alpha)))
        ft_indptr, ft_rows, ft_perm = itf.transposed()
        seed = (model.sampler.seed if self.device_sampler else 0) & (2 ** 64 - 1)
        lib = N.load()
        with torch.no_grad():
            rc = lib.trec_fit_step_coop(
                N.ptr(wu), N.ptr(mu), N.ptr(vu), N.ptr(wi), N.ptr(mi), N.ptr(vi), N.ptr(bu), N.ptr(bum), N.ptr(buv), N.ptr(bi),
                N.ptr(bim), N.ptr(biv), N.ptr(itf.indptr), N.ptr(itf.indices), N.ptr(itf.values), N.ptr(ft_indptr), N.ptr(ft_rows),
                N.ptr(ft_perm), N.ptr(inter.indptr), N.ptr(inter.x_item32), N.ptr(inter.pos_slot), N.ptr(self.weight), N.ptr(samples),
                inter.shape[0], inter.shape[1], itf.shape[1], int(model.n_components), self.S, int(inter.max_row_nnz),
                int(getattr(inter, 'user_base', 0)), seed, int(model._sample_step) & 0xFFFFFFFF, lr_t, ADAM_BETA1, ADAM_BETA2,
                ADAM_EPSILON, l2, N.ptr(self.ws), int(self.ws.numel()), N.ptr(self.loss), N.ptr(self.pred), N.stream())
        if rc == 3:                                             # TREC_ERR_UNSUPPORTED: no cooperative launch on this device / occupancy
            model._sample_step -= 1
            self.ok = False
            return False
        if rc != 0:
            raise RuntimeError("trec_fit_step_coop failed (code %d): %s" % (rc, lib.trec_last_error().decode()))
        model._opt_step = t
        model._schedule_mirror = None                           # (the device-side schedule of graph replays is stale now)
        if want_stats:
            with torch.no_grad():
                wr = float(sum(0.5 * float((v[n].detach() ** 2).sum()) for n in self.names))
            return self.loss.detach(), self.pred.detach(), wr
        return None, None, None


def _merge_taste_topk(per_taste, k):
    """Top-k of max-over-tastes from the per-taste top-k lists.  Exact, ties included: rounding is monotone, so adding
    the biases before the max gives the same floats as after it, and an item of the collapsed top-k is in the top-k
    list of the taste that attains its maximum (everything ahead of it in that list is ahead of it in the collapsed
    order too).  The lists are [n_users, T * k] -- a few KB per user -- so the merge is index plumbing: order by item
    id, keep the best copy of each item, then order by (score desc, item asc)."""
    v = torch.cat([p[0] for p in per_taste], dim=1)
    i = torch.cat([p[1] for p in per_taste], dim=1)
    o = torch.sort(v, dim=1, descending=True, stable=True).indices
    v, i = torch.gather(v, 1, o), torch.gather(i, 1, o)
    key = torch.where(i < 0, torch.full_like(i, 2 ** 31 - 1), i)
    o = torch.sort(key, dim=1, stable=True).indices           # item asc; equal items keep score-desc order
    v, i, key = torch.gather(v, 1, o), torch.gather(i, 1, o), torch.gather(key, 1, o)
    dup = torch.zeros_like(i, dtype=torch.bool)
    dup[:, 1:] = key[:, 1:] == key[:, :-1]
    v = torch.where(dup, torch.full_like(v, float('-inf')), v)
    i = torch.where(dup | (i < 0), torch.full_like(i, -1), i)
    key = torch.where(i < 0, torch.full_like(i, 2 ** 31 - 1), i)
    o = torch.sort(key, dim=1, stable=True).indices           # duplicates (now -1) to the back, item asc kept
    v, i = torch.gather(v, 1, o), torch.gather(i, 1, o)
    o = torch.sort(v, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(v, 1, o).contiguous(), torch.gather(i, 1, o).contiguous()


class _DenseDot(torch.autograd.Function):
    """Differentiable dense scores for is_dense losses: forward = K2 STORE kernel, backward = two fp32 MFMA GEMMs."""

    @staticmethod
    def forward(ctx, u, v):
        ctx.save_for_backward(u, v)
        return ops.dense_scores(u, v, ops.DTYPE_F32)

    @staticmethod
    def backward(ctx, g):
        u, v = ctx.saved_tensors
        g = g.contiguous()
        return ops.gemm_raw(g, v), ops.gemm_raw(g, u, trans_a=True)


def _differentiable_dense(graph, user_repr, item_repr):
    if graph.engine_mode == ops.MODE_DOT:
        if graph.engine_normalize:
            user_repr, item_repr = ops.l2_normalize_rows(user_repr), ops.l2_normalize_rows(item_repr)
        return _DenseDot.apply(user_repr, item_repr)
    # euclidean: r_u - 2 u.i + r_i through the differentiable dot (prediction_graphs.py:84-100)
    r_u = (user_repr ** 2).sum(dim=1, keepdim=True)
    r_i = (item_repr ** 2).sum(dim=1, keepdim=True)
    dist = (r_u - 2.0 * _DenseDot.apply(user_repr, item_repr)) + r_i.t()
    return -1.0 * torch.sqrt(torch.clamp(dist, min=1e-16))
