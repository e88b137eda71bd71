"""
Device-resident sparse inputs.

The reference turns every scipy.sparse input into a COO 5-tuple (rows int64, cols int64, values float32, d0, d1)
(tensorrec/input_utils.py:22-40) and rebuilds ``tf.SparseTensor``s from it (tensorrec/tensorrec.py:285-295).
Here the same data lives in HBM as CSR (row pointers int64, column indices int32, values fp32) plus -- built
lazily, once -- the CSR of the transpose, which turns the backward SpMM (a dense [F, d] weight gradient in TF) into
the same gather kernel.  Entry order is row-major (what ``sp.coo_matrix(csr)`` yields), which is the order of
``tf_interactions.values`` / ``tf_prediction_serial`` in the reference for CSR inputs.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


class SparseFeatures(object):
    """A [n_rows, n_cols] feature matrix on the device (stands in for the ``tf.SparseTensor`` handed to
    ``connect_representation_graph``)."""

    def __init__(self, matrix, device="cuda"):
        if not sp.issparse(matrix):
            raise ValueError("Input must be a scipy sparse matrix")
        m = sp.csr_matrix(matrix)
        if not m.has_sorted_indices:
            m = m.sorted_indices()
        m.sum_duplicates()
        if m.shape[0] >= 2 ** 31 or m.shape[1] >= 2 ** 31:
            raise ValueError("dimensions must fit int32")
        self._host = m
        self.shape = (int(m.shape[0]), int(m.shape[1]))
        self.nnz = int(m.nnz)
        self.device = torch.device(device)
        self.indptr = _dev(m.indptr.astype(np.int64), self.device)
        self.indices = _dev(m.indices.astype(np.int32), self.device)
        self.values = _dev(m.data.astype(np.float32), self.device)
        # longest row / column: the gathers switch to their chunked form above ops.SPLIT_T non-zeros
        self.max_row_nnz = int(np.diff(m.indptr).max()) if m.shape[0] and m.nnz else 0
        self.max_col_nnz = int(np.bincount(m.indices, minlength=1).max()) if m.nnz else 0
        # exactly one non-zero in every row (identity / indicator features): K1 then skips the row pointer
        self.one_per_row = bool(m.shape[0] > 0 and m.nnz == m.shape[0] and
                                np.array_equal(m.indptr, np.arange(m.shape[0] + 1)))
        # the identity matrix itself (sp.identity user / item features, BASELINE.json configs[2]): X . W IS W -- the linear
        # representation then aliases the weight table instead of copying it through K1 (ops.sparse_dense_matmul)
        self.is_identity = bool(self.one_per_row and m.shape[0] == m.shape[1] and
                                np.array_equal(m.indices, np.arange(m.shape[0], dtype=m.indices.dtype)) and
                                bool(np.all(m.data == 1.0)))
        # column support [min col, max col + 1) of the non-zeros: the rows of a weight table this matrix can touch (the
        # data-parallel fit skips the gradient exchange of tables whose supports do not overlap between ranks: sharding.py)
        self.col_range = (int(m.indices.min()), int(m.indices.max()) + 1) if m.nnz else (0, 0)
        self.role = None                # "user" / "item": set by TensorRec.fit_partial
        self._t = None

    @property
    def dense_shape(self):
        return self.shape

    def transposed(self):
        """(indptr_t int64 [n_cols+1], rows_t int32 [nnz], perm_t int32 [nnz]): CSR of X^T whose entry j carries the
        value ``values[perm_t[j]]`` -- no second copy of the values is needed."""
        if self._t is None:
            m = self._host
            tagged = sp.csr_matrix((np.arange(1, m.nnz + 1, dtype=np.int64), m.indices, m.indptr), shape=m.shape)
            t = sp.csr_matrix(tagged.T)
            if not t.has_sorted_indices:
                t = t.sorted_indices()
            self._t = (_dev(t.indptr.astype(np.int64), self.device), _dev(t.indices.astype(np.int32), self.device),
                       _dev((t.data - 1).astype(np.int32), self.device))
        return self._t

    def row_slice(self, start, end):
        return SparseFeatures(self._host[start:end], self.device)

    def to_scipy(self):
        return self._host


class Interactions(object):
    """The [n_users, n_items] interaction matrix on the device, in the forms the loss graphs need:

    * ``x_user`` / ``x_item`` (int64) and ``values``: the serial (COO, row-major) view -- what
      ``split_sparse_tensor_indices`` (recommendation_graphs.py:22-30) and ``tf_interactions.values`` give;
    * ``indptr``: CSR row pointers over users (WMRB walks a user's interactions);
    * ``pos_slot`` / ``n_positive``: position of each interaction inside ``tf.boolean_mask(..., values > 0)``
      (loss_graphs.py:155-161), -1 for non-positive ones.

    ``shape`` follows the reference: taken from the FEATURE matrices, not from the interactions' own shape
    (tensorrec.py:294-295)."""

    def __init__(self, matrix, n_users, n_items, device="cuda"):
        if not sp.issparse(matrix):
            raise ValueError("Input must be a scipy sparse matrix")
        m = sp.csr_matrix(matrix)
        if not m.has_sorted_indices:
            m = m.sorted_indices()
        m.sum_duplicates()
        if m.shape[0] > n_users or m.shape[1] > n_items:
            raise ValueError("interactions of shape %s do not fit [n_users=%d, n_items=%d] given by the features"
                             % (m.shape, n_users, n_items))
        self.device = torch.device(device)
        self.shape = (int(n_users), int(n_items))
        self.dense_shape = self.shape
        self.nnz = int(m.nnz)
        indptr = np.zeros(n_users + 1, np.int64)
        indptr[: m.shape[0] + 1] = m.indptr
        indptr[m.shape[0] + 1:] = m.indptr[-1]
        coo_rows = np.repeat(np.arange(m.shape[0], dtype=np.int64), np.diff(m.indptr))
        vals = m.data.astype(np.float32)
        pos = vals > 0.0
        pos_slot = np.full(m.nnz, -1, np.int32)
        pos_slot[pos] = np.arange(int(pos.sum()), dtype=np.int32)
        self.n_positive = int(pos.sum())
        self.max_row_nnz = int(np.diff(m.indptr).max()) if m.shape[0] and m.nnz else 0
        self.max_col_nnz = int(np.bincount(m.indices, minlength=1).max()) if m.nnz else 0     # the most popular item
        self.indptr = _dev(indptr, self.device)
        self.x_user = _dev(coo_rows, self.device)
        self.x_item = _dev(m.indices.astype(np.int64), self.device)
        self.x_user32 = self.x_user.to(torch.int32)
        self.x_item32 = self.x_item.to(torch.int32)
        self.values = _dev(vals, self.device)
        self.pos_slot = _dev(pos_slot, self.device)
        self._host = m
        self._balanced_weight = None
        self._t = None

    def transposed(self):
        """(indptr_t int64 [n_items+1], users_t int32 [nnz], perm_t int32 [nnz]): the interactions grouped by item;
        entry j of the transposed structure is serial interaction ``perm_t[j]``.  Lets the item-side gradient of the
        serial predictions be a deterministic segmented gather (K1) instead of atomics."""
        if self._t is None:
            m = self._host
            tagged = sp.csr_matrix((np.arange(1, m.nnz + 1, dtype=np.int64), m.indices, m.indptr), shape=m.shape)
            t = sp.csr_matrix(tagged.T)
            if not t.has_sorted_indices:
                t = t.sorted_indices()
            indptr_t = np.zeros(self.shape[1] + 1, np.int64)
            indptr_t[: t.shape[0] + 1] = t.indptr
            indptr_t[t.shape[0] + 1:] = t.indptr[-1]
            self._t = (_dev(indptr_t, self.device), _dev(t.indices.astype(np.int32), self.device),
                       _dev((t.data - 1).astype(np.int32), self.device))
        return self._t

    @property
    def indices(self):
        """[nnz, 2] int64, like ``tf.SparseTensor.indices``."""
        return torch.stack([self.x_user, self.x_item], dim=1)

    def balanced_weight(self):
        """value_p / (sum of positive values of p's item)  -- loss_graphs.py:197-202, 222-224; 0 for non-positives.
        Under a user-sharded (data-parallel) fit the trainer sets ``self.dp_group = (True, group)`` and the per-item
        sums are all-reduced over THAT group so they cover every user; in every other situation -- including item-sharded
        inference processes that each fit the full data -- the sums are local.  Cached per mode."""
        active, group = getattr(self, "dp_group", (False, None))
        key = (bool(active), id(group) if active else None)
        if self._balanced_weight is None or self._balanced_weight[0] != key:
            m = self._host
            vals = m.data.astype(np.float32)
            pos = vals > 0.0
            per_item = np.zeros(self.shape[1], np.float32)
            np.add.at(per_item, m.indices[pos], vals[pos])
            if active:
                import torch.distributed as dist
                t = _dev(per_item, self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                per_item = t.cpu().numpy()
            w = np.zeros(m.nnz, np.float32)
            w[pos] = vals[pos] / per_item[m.indices[pos]]
            self._balanced_weight = (key, _dev(w, self.device))
        return self._balanced_weight[1]

    def to_scipy(self):
        return self._host


class PairIndex(torch.Tensor):
    """An int64 index tensor (what ``tf_x_user`` / ``tf_x_item`` are in the reference) that also carries an int32
    copy for the kernels and, for sampled pairs, the fact that users are implicit (pair p belongs to user
    p // pairs_per_user).  Custom prediction graphs can index with it like any LongTensor."""

    @staticmethod
    def make(idx64, idx32=None, pairs_per_user=0, interactions=None):
        t = idx64.as_subclass(PairIndex)
        t.idx32 = idx32 if idx32 is not None else idx64.to(torch.int32)
        t.pairs_per_user = int(pairs_per_user)
        t.interactions = interactions          # Interactions whose serial order these indices follow (or None)
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # behave as a plain tensor in every torch op (results are plain tensors)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))
