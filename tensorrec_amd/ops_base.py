"""
ops_base -- the kernels of the path that are NOT the user x item contraction: K1 (sparse features x weights, its backward and
epilogues), the fp32 MFMA GEMM, K3 pair scores, the device grouping of pair lists by item, K9 mixture of tastes, K6 losses (WMRB
fused and unfused, RMSE, dense / separation), K4 ranks of a score slab, K7 sampler, K8 Adam.  Thin autograd wrappers over
libtensorrec_hip.so (tensorrec_amd/_native.py); the public module is tensorrec_amd.ops, which re-exports everything here.
"""
from __future__ import annotations

import torch

from . import _native as N
from .sparse import SparseFeatures, Interactions, PairIndex

DTYPE_F32, DTYPE_BF16 = 0, 1

# bench.py sets this to a list to collect (start, end) HIP events around every launch of the named kernels on the
# launching stream (the kernels run on torch's current stream, so torch.cuda.Event brackets exactly them)
KERNEL_EVENTS = None


class _timed(object):
    """Brackets a launch (or a group of launches) with HIP events; brackets nested inside another one are not recorded,
    so a composite step (e.g. the exact fallback of the filtered top-k) does not pollute its inner kernels' statistics."""
    depth = 0

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = KERNEL_EVENTS is not None and _timed.depth == 0
        _timed.depth += 1
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        _timed.depth -= 1
        if self.on and KERNEL_EVENTS is not None:
            self.e.record()
            KERNEL_EVENTS.append((self.name, self.s, self.e))
        return False
MODE_DOT, MODE_EUCLIDEAN = 0, 1
EPI_NONE, EPI_L2NORM, EPI_BIAS_RELU, EPI_ROWSUM = 0, 1, 2, 3


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ K1
def spmm_raw(indptr, indices, values, perm, n_rows, nnz, w, col_bias=None, epilogue=EPI_NONE, accumulate=False,
             out=None, want_inv=False, one_per_row=False):
    w = _f32c(w)
    d = w.shape[1]
    if out is None:
        out = torch.empty((n_rows, d), dtype=torch.float32, device=w.device)
    if one_per_row and perm is None and epilogue == EPI_NONE and not accumulate and not want_inv and d % 4 == 0 \
            and d <= 1024 and nnz == n_rows and N.load().trec_get_tuning(b"spmm_one_per_row", 1):
        # identity / indicator features (SparseFeatures.one_per_row): the row pointer is the identity and is not read
        with _timed("spmm_csr"):
            N.call("trec_spmm_one_per_row", N.ptr(indices), N.ptr(values), n_rows, N.ptr(w), d, N.ptr(out))
        return out
    inv = torch.empty((n_rows,), dtype=torch.float32, device=w.device) if want_inv else None
    with _timed("spmm_csr"):
        N.call("trec_spmm_csr", N.ptr(indptr), N.ptr(indices), N.ptr(values), N.ptr(perm), n_rows, nnz, N.ptr(w), d,
               N.ptr(col_bias), epilogue, 1 if accumulate else 0, N.ptr(out), N.ptr(inv))
    return (out, inv) if want_inv else out


# rows longer than this are cut into chunks by the split gathers (csrc/spmm_split.hip: SPLIT_T_DEFAULT)
SPLIT_T = 2048
# uniform samples: a bucket of mean m holds at most ~m + 4.5 sqrt(m) pairs; above this mean some bucket passes SPLIT_T
_SPLIT_MEAN = 1700


def _split_workspace(nnz, d, dev):
    nbytes = N.query("trec_csr_split_workspace_bytes", int(nnz), int(d))
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def spmm_split(indptr, indices, values, perm, n_rows, nnz, w, own=None, accumulate=False, out=None, want_rowsum=False,
               packed=None):
    """K1 for skewed row lengths (trec_spmm_csr_split): rows of more than SPLIT_T non-zeros are summed as chunks by the
    whole chip; ``own``: gather (own[row] - w[col]) -- the Euclidean pair gradient."""
    w = _f32c(w)
    d = w.shape[1]
    if out is None:
        out = torch.empty((n_rows, d), dtype=torch.float32, device=w.device)
    rowsum = torch.empty((n_rows,), dtype=torch.float32, device=w.device) if want_rowsum is True else \
        (want_rowsum if isinstance(want_rowsum, torch.Tensor) else None)
    ws, nbytes = _split_workspace(nnz, d, w.device)
    own = _f32c(own) if own is not None else None
    with _timed("spmm_csr_split"):
        N.call("trec_spmm_csr_split", N.ptr(indptr), N.ptr(indices), N.ptr(values), N.ptr(perm), N.ptr(packed), n_rows,
               nnz, N.ptr(w), d, N.ptr(own), 1 if accumulate else 0, N.ptr(out), N.ptr(rowsum), N.ptr(ws), nbytes)
    return (out, rowsum) if want_rowsum is True else out


def spmv_raw(indptr, indices, values, perm, n_rows, nnz, beta, long_rows=False):
    """out[r] = sum_j values[j] * beta[indices[j]] (beta None: plain segment sums); ``long_rows``: some row may exceed
    SPLIT_T non-zeros -> the chunked form."""
    out = torch.empty((n_rows,), dtype=torch.float32, device=indptr.device)
    if long_rows:
        ws, nbytes = _split_workspace(nnz, 1, indptr.device)
        N.call("trec_spmv_csr_split", N.ptr(indptr), N.ptr(indices), N.ptr(values), N.ptr(perm), n_rows, nnz,
               N.ptr(beta), N.ptr(out), N.ptr(ws), nbytes)
    else:
        N.call("trec_spmv_csr", N.ptr(indptr), N.ptr(indices), N.ptr(values), N.ptr(perm), n_rows, N.ptr(beta),
               N.ptr(out))
    return out


def _split_ok(d):
    """widths the chunked gathers cover: float4 lanes up to 1024 columns, single-column lanes (any d) up to 256"""
    return (d % 4 == 0 and d <= 1024) or d <= 256


def _prefer_split(d, nnz):
    """d % 4 != 0 has no float4 kernel: K1's fallback walks a row with one load in flight per thread, the chunked form
    keeps four -- worth its three extra launches on large gathers"""
    return d % 4 != 0 and d <= 256 and nnz >= 65536


def _sampled_buckets_long(n_pairs, n_items):
    """pairs grouped by sampled item on the device: bucket sizes are not known to the host.  Uniform samples stay under
    SPLIT_T up to a mean of _SPLIT_MEAN; large problems take the split form regardless (three small extra launches) so
    that a skewed custom sampler does not serialise on its popular items."""
    return n_pairs >= (1 << 20) or n_pairs > n_items * _SPLIT_MEAN


def _spmm_t(feats: SparseFeatures, dout):
    """dW[F, d] = X^T . dOut -- the same gather kernel on the transposed CSR (chunked where a feature column is long:
    indicator columns of side features hold thousands of rows)."""
    indptr_t, rows_t, perm_t = feats.transposed()
    d = dout.shape[1]
    if (feats.max_col_nnz > SPLIT_T or _prefer_split(d, feats.nnz)) and _split_ok(d):
        return spmm_split(indptr_t, rows_t, feats.values, perm_t, feats.shape[1], feats.nnz, dout)
    return spmm_raw(indptr_t, rows_t, feats.values, perm_t, feats.shape[1], feats.nnz, dout)


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, feats):
        ctx.feats = feats
        return spmm_raw(feats.indptr, feats.indices, feats.values, None, feats.shape[0], feats.nnz, w,
                        one_per_row=getattr(feats, 'one_per_row', False))

    @staticmethod
    def backward(ctx, dout):
        return _spmm_t(ctx.feats, _f32c(dout)), None


class _SpMMNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, feats):
        y, inv = spmm_raw(feats.indptr, feats.indices, feats.values, None, feats.shape[0], feats.nnz, w,
                          epilogue=EPI_L2NORM, want_inv=True)
        ctx.feats = feats
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        N.call("trec_row_l2norm_bwd", N.ptr(y), N.ptr(inv), N.ptr(_f32c(dy)), y.shape[0], y.shape[1], N.ptr(dx))
        return _spmm_t(ctx.feats, dx), None


class _SpMMBiasRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, bias, feats):
        b = _f32c(bias).reshape(-1)
        out = spmm_raw(feats.indptr, feats.indices, feats.values, None, feats.shape[0], feats.nnz, w, col_bias=b,
                       epilogue=EPI_BIAS_RELU)
        ctx.feats = feats
        ctx.bias_shape = bias.shape
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dpre = torch.empty_like(out)
        N.call("trec_relu_bwd", N.ptr(out), N.ptr(_f32c(dout)), out.numel(), N.ptr(dpre))
        return _spmm_t(ctx.feats, dpre), colsum(dpre).reshape(ctx.bias_shape), None


def colsum(x):
    """Column sums of a [n_rows, d] matrix (gradient of a broadcast bias): row slices in parallel, added in slice order."""
    x = _f32c(x)
    n_rows, d = x.shape
    out = torch.empty((d,), dtype=torch.float32, device=x.device)
    n_slices = max(1, min(512, n_rows // 256))
    ws = torch.empty((n_slices, d), dtype=torch.float32, device=x.device) if n_slices > 1 else None
    with _timed("colsum"):
        N.call("trec_colsum", N.ptr(x), n_rows, d, N.ptr(out), N.ptr(ws), n_slices)
    return out


def _note_row_support(w, feats):
    """A trainable table multiplied with a sparse feature matrix receives gradient only in the rows that matrix has columns
    for.  The table remembers WHICH matrices it met this fit call (``w._trec_feats``); the data-parallel fit turns them into
    the rows this rank can touch and skips the gradient exchange of tables whose row supports are rank-disjoint
    (TensorRec._dp_make_plan, sharding.plan_gradient_exchange).  Any other use of a table leaves it unmarked: its gradient is
    exchanged in full."""
    if isinstance(w, torch.Tensor) and w.requires_grad and w.is_leaf:
        met = getattr(w, "_trec_feats", None)
        if met is None:
            met = w._trec_feats = set()
        met.add(id(feats))


def sparse_dense_matmul(feats: SparseFeatures, w):
    """tf.sparse_tensor_dense_matmul(features, w)"""
    _note_row_support(w, feats)
    if getattr(feats, "is_identity", False) and w.dim() == 2 and w.shape[0] == feats.shape[1] and w.dtype == torch.float32 \
            and w.is_contiguous() and N.load().trec_get_tuning(b"identity_alias", 1) != 0:
        # identity features: the product IS the table (1.0 * w + 0 exactly).  A view, not a copy: 0.33 ms of K1 forward and
        # 0.46 ms of its gather-copy backward per epoch of the 1M x 1M fit; the gradient of the view is the table's gradient
        out = w.view(w.shape)
        out._trec_alias_of = w
        return out
    return _SpMM.apply(w, feats)


def accumulate_grads(tensors, grads):
    """torch.autograd.backward(tensors, grads) for the hand-made gradients of the fused steps -- except that a tensor which is
    a plain alias of a leaf table (identity features, sparse_dense_matmul) hands its gradient to the table directly:
    AccumulateGrad would clone it (the caller still holds a reference), 512 MB per side at 1M x 128."""
    rest_t, rest_g = [], []
    for t, g in zip(tensors, grads):
        if g is None or not t.requires_grad:
            continue
        leaf = getattr(t, "_trec_alias_of", None)
        if leaf is not None and leaf.is_leaf and leaf.grad is None and g.shape == leaf.shape and g.dtype == leaf.dtype:
            leaf.grad = g
        else:
            rest_t.append(t)
            rest_g.append(g)
    if rest_t:
        torch.autograd.backward(rest_t, rest_g)


def sparse_dense_matmul_l2norm(feats, w):
    _note_row_support(w, feats)
    return _SpMMNorm.apply(w, feats)


def sparse_dense_matmul_bias_relu(feats, w, bias):
    _note_row_support(w, feats)
    return _SpMMBiasRelu.apply(w, bias, feats)


class _SpMV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, beta, feats):
        ctx.feats = feats
        ctx.beta_shape = beta.shape
        return spmv_raw(feats.indptr, feats.indices, feats.values, None, feats.shape[0], feats.nnz,
                        _f32c(beta).reshape(-1), feats.max_row_nnz > SPLIT_T)

    @staticmethod
    def backward(ctx, dout):
        feats = ctx.feats
        indptr_t, rows_t, perm_t = feats.transposed()
        dbeta = spmv_raw(indptr_t, rows_t, feats.values, perm_t, feats.shape[1], feats.nnz, _f32c(dout),
                         feats.max_col_nnz > SPLIT_T)
        return dbeta.reshape(ctx.beta_shape), None


def sparse_matvec(feats, beta):
    _note_row_support(beta, feats)
    return _SpMV.apply(beta, feats)


def sparse_to_dense(feats: SparseFeatures):
    out = torch.empty(feats.shape, dtype=torch.float32, device=feats.device)
    N.call("trec_csr_to_dense", N.ptr(feats.indptr), N.ptr(feats.indices), N.ptr(feats.values), feats.shape[0],
           feats.shape[1], N.ptr(out))
    return out


class _RowL2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        y = torch.empty_like(x)
        inv = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
        N.call("trec_row_l2norm_fwd", N.ptr(x), x.shape[0], x.shape[1], N.ptr(y), N.ptr(inv))
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        N.call("trec_row_l2norm_bwd", N.ptr(y), N.ptr(inv), N.ptr(_f32c(dy)), y.shape[0], y.shape[1], N.ptr(dx))
        return dx


def l2_normalize_rows(x):
    """tf.nn.l2_normalize(x, 1)"""
    return _RowL2Norm.apply(x)


def gemm_raw(a, b, trans_a=False, trans_b=False, split_bf16=False, a_colsum=False):
    """fp32 C = op(a) . op(b).  ``split_bf16``: both operands as two bf16 terms on bf16 MFMA (three products, fp32 accumulation:
    ~1e-5 relative) -- for gradients held to 1e-4, never for a value the oracle's fmaf chain is compared with.  ``a_colsum`` (with
    split_bf16 and trans_a): also returns the column sums of the stored ``a`` -- (C, colsum) -- taken while ``a`` is staged."""
    a, b = _f32c(a), _f32c(b)
    m = a.shape[1] if trans_a else a.shape[0]
    k = a.shape[0] if trans_a else a.shape[1]
    n = b.shape[0] if trans_b else b.shape[1]
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    # few output tiles and a long K (dW2 = Relu^T . dOut: [relu_size, n_components] over all users): split K so that the
    # launch has ~1024 workgroups; the slices are added in order (deterministic)
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    splits = max(1, min(1024 // max(tiles, 1), k // 512)) if tiles < 512 else 1
    ws = torch.empty((splits, m, n), dtype=torch.float32, device=a.device) if splits > 1 else None
    if split_bf16:
        k_per = -(-(-(-k // splits)) // 64) * 64                           # (the slices the entry point makes: whole 64-wide slabs)
        n_slices = max(1, -(-k // max(k_per, 64)))
        parts = torch.empty((n_slices, m), dtype=torch.float32, device=a.device) if (a_colsum and trans_a) else None
        N.call("trec_gemm_f32_split_bf16", 1 if trans_a else 0, 1 if trans_b else 0, m, n, k, N.ptr(a), a.shape[1], N.ptr(b),
               b.shape[1], N.ptr(c), n, 0, N.ptr(ws), splits, N.ptr(parts))
        if a_colsum:
            return c, (parts.sum(dim=0) if parts is not None else colsum(a))
        return c
    N.call("trec_gemm_f32", 1 if trans_a else 0, 1 if trans_b else 0, m, n, k, N.ptr(a), a.shape[1], N.ptr(b),
           b.shape[1], N.ptr(c), n, 0, N.ptr(ws), splits)
    return (c, colsum(a)) if a_colsum else c


class _MatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm_raw(a, b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = _f32c(dc)
        return gemm_raw(dc, b, trans_b=True), gemm_raw(a, dc, trans_a=True)


def matmul(a, b):
    """tf.matmul(a, b) for the dense layer of ReLURepresentationGraph"""
    return _MatMul.apply(a, b)


# ------------------------------------------------------------------------------------------------ K3
def _idx32(x):
    if isinstance(x, PairIndex):
        return x.idx32, x.pairs_per_user
    return x.to(torch.int32).contiguous(), 0


class _PairScore(torch.autograd.Function):
    """Forward: K3.  Backward, for structured pair lists (a user's pairs are consecutive):
      * user side -- pairs are user-major (CSR over users: interactions, or S consecutive samples per user), so
        dU = G . V is the K1 gather kernel with the pair gradients as values: deterministic, no atomics;
      * item side -- dV = G^T . U is the same kernel on the pairs grouped by item: interactions carry that structure,
        sampled pairs (random items) are grouped by a counting sort on the device;
      * Euclidean pairs: the values are c_p = -g_p / sqrt(D_p) (trec_pair_euclid_coef) and every gathered row is
        (own - other), i.e. dU[u] = sum_p c_p (U[u] - V[i_p]), dV[i] = sum_p c_p (V[i] - U[u_p]);
      * rows longer than SPLIT_T pairs (popular items of Zipf-shaped data) are summed as chunks (spmm_split.hip).
    Unstructured index tensors use the atomic kernel for both sides."""

    @staticmethod
    def forward(ctx, u, v, ub, ib, xu32, xi32, pairs_per_user, mode, inter):
        u, v = _f32c(u), _f32c(v)
        n_pairs = xi32.numel()
        out = torch.empty((n_pairs,), dtype=torch.float32, device=u.device)
        # Euclidean pairs keep their squared distances: the backward coefficients -g / sqrt(D) need no second gather
        keep = mode == MODE_EUCLIDEAN and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        sqdist = torch.empty((n_pairs,), dtype=torch.float32, device=u.device) if keep else None
        N.call("trec_pair_score_fwd", N.ptr(u), N.ptr(v), N.ptr(xu32), N.ptr(xi32), n_pairs, pairs_per_user, u.shape[1],
               mode, N.ptr(ub), N.ptr(ib), N.ptr(out), N.ptr(sqdist))
        ctx.sqdist = sqdist
        ctx.save_for_backward(u, v)
        ctx.meta = (xu32, xi32, pairs_per_user, mode, ub is not None, ib is not None, inter)
        return out

    @staticmethod
    def backward(ctx, g):
        u, v = ctx.saved_tensors
        xu32, xi32, ppu, mode, has_ub, has_ib, inter = ctx.meta
        g = _f32c(g)
        n_pairs = xi32.numel()
        n_users, n_items, dev = u.shape[0], v.shape[0], u.device
        d = u.shape[1]
        structured = (ppu > 0 or inter is not None) and (mode == MODE_DOT or _split_ok(d))
        if not structured:
            du, dv = torch.zeros_like(u), torch.zeros_like(v)
            dub = torch.zeros((n_users,), dtype=torch.float32, device=dev) if has_ub else None
            dib = torch.zeros((n_items,), dtype=torch.float32, device=dev) if has_ib else None
            N.call("trec_pair_score_bwd", N.ptr(u), N.ptr(v), N.ptr(xu32), N.ptr(xi32), N.ptr(g), n_pairs, ppu,
                   u.shape[1], mode, N.ptr(du), N.ptr(dv), N.ptr(dub), N.ptr(dib))
            return du, dv, dub, dib, None, None, None, None, None
        euclid = mode == MODE_EUCLIDEAN
        vals = g
        if euclid:       # dU[u] = sum_p c_p (U[u] - V[i_p]),  dV[i] = sum_p c_p (V[i] - U[u_p])
            vals = torch.empty_like(g)
            N.call("trec_pair_euclid_coef", N.ptr(u), N.ptr(v), N.ptr(xu32), N.ptr(xi32), N.ptr(g), N.ptr(ctx.sqdist),
                   n_pairs, ppu, d, N.ptr(vals))
        can_split = _split_ok(d)
        prefer = _prefer_split(d, n_pairs)
        # ---- user side: segmented gather over each user's pairs (K1 with values = g)
        if inter is not None:
            indptr_u, long_u = inter.indptr, inter.max_row_nnz > SPLIT_T
        else:
            indptr_u = torch.arange(0, (n_users + 1) * ppu, ppu, dtype=torch.int64, device=dev)
            long_u = ppu > SPLIT_T
        if euclid or ((long_u or prefer) and can_split):
            du = spmm_split(indptr_u, xi32, vals, None, n_users, n_pairs, v, own=u if euclid else None)
        else:
            du = spmm_raw(indptr_u, xi32, vals, None, n_users, n_pairs, v)
        dub = spmv_raw(indptr_u, xi32, g, None, n_users, n_pairs, None, long_u) if has_ub else None
        # ---- item side: interactions carry their transposed structure; sampled pairs (random items, no structure) are
        # grouped by item on the device (counting sort) -- then the same segmented gather instead of n_pairs * d atomics
        if inter is not None:
            indptr_t, users_t, perm_t = inter.transposed()
            long_i = inter.max_col_nnz > SPLIT_T
        else:
            indptr_t, users_t, perm_t = group_pairs_by_item(xu32, xi32, ppu, n_items)
            long_i = _sampled_buckets_long(n_pairs, n_items)
        if euclid or ((long_i or prefer) and can_split):
            dv = spmm_split(indptr_t, users_t, vals, perm_t, n_items, n_pairs, u, own=v if euclid else None)
        else:
            dv = spmm_raw(indptr_t, users_t, vals, perm_t, n_items, n_pairs, u)
        dib = spmv_raw(indptr_t, users_t, g, perm_t, n_items, n_pairs, None, long_i) if has_ib else None
        return du, dv, dub, dib, None, None, None, None, None


import threading as _threading


class _Local(_threading.local):
    deterministic_grouping = False
    loss_group = None          # (process group,) while a user-sharded fit step builds its loss: scalar losses span ALL shards


_LOCAL = _Local()       # per THREAD: two models fitting in different threads must not switch each other's grouping mode


class deterministic_grouping(object):
    """``with ops.deterministic_grouping(flag):`` -- group_pairs_by_item uses the stable (bit-reproducible) grouping inside the
    block, in this thread only; the previous mode returns on exit (TensorRec(deterministic=True) wraps its fit calls in it)."""

    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        self.prev = _LOCAL.deterministic_grouping
        _LOCAL.deterministic_grouping = self.flag
        return self

    def __exit__(self, *exc):
        _LOCAL.deterministic_grouping = self.prev
        return False


class scalar_loss_group(object):
    """``with ops.scalar_loss_group(group):`` -- inside, the scalar losses (RMSE, RMSEDense, Separation, SeparationDense:
    loss_graphs.py:58-134) are those of the UNION of every rank's interactions: their sums are all-reduced over ``group``
    (a collective: every rank of the group must build the same loss) and the backward pass differentiates the global scalar
    with respect to this rank's predictions.  TensorRec's data-parallel step wraps its loss construction in it."""

    def __init__(self, group, active=True):
        self.value = (group,) if active else None

    def __enter__(self):
        self.prev = _LOCAL.loss_group
        _LOCAL.loss_group = self.value
        return self

    def __exit__(self, *exc):
        _LOCAL.loss_group = self.prev
        return False


def _loss_all_reduce(t):
    """SUM over the ranks of the current scalar_loss_group, in place (float64 / float32 device tensor)."""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_LOCAL.loss_group[0])
    return t


def group_pairs_by_item(xu32, xi32, pairs_per_user, n_items, workspace_with_counts=None, ranks=None, values=None):
    """(indptr_t int64 [n_items+1], users_t int32 [n_pairs], perm_t int32 [n_pairs]) for a pair list, built on the
    device; the order of pairs inside an item's bucket is not fixed (atomic slot assignment).
    ``workspace_with_counts``: an int32 [2 * n_items] workspace whose first half already holds the histogram of the
    items (counted by the kernel that consumed the pairs) -- the histogram pass is then skipped; ``ranks`` (with it): the
    values those histogram atomics returned, which makes the fill pass atomic-free; ``values`` (with ranks): per-pair
    values to carry along -- the result is then (indptr_t, entries int32 [n_pairs, 2] = {user, value bits}, None): one
    8-byte scattered store per pair, consumed by trec_spmm_csr_packed."""
    dev = xi32.device
    n_pairs = xi32.numel()
    if _LOCAL.deterministic_grouping and n_pairs:
        # bit-reproducible fits (TensorRec(deterministic=True)): a STABLE sort by item keeps the pairs of a bucket in pair
        # order, so the fp32 sums over a bucket are added in the same order every run (the counting sort below orders a
        # bucket by atomic arrival).  Negative keys sort to the front and fall before indptr[0].
        order = torch.sort(xi32, stable=True).indices
        sorted_keys = xi32[order]
        indptr_t = torch.searchsorted(sorted_keys, torch.arange(n_items + 1, dtype=torch.int32, device=dev)).to(torch.int64)
        users = xu32[order] if xu32 is not None else torch.div(order, pairs_per_user, rounding_mode="floor").to(torch.int32)
        if values is not None and ranks is not None:
            entries = torch.stack([users.to(torch.int32), values[order].contiguous().view(torch.int32)], dim=1).contiguous()
            return indptr_t, entries, None
        return indptr_t, users.to(torch.int32).contiguous(), order.to(torch.int32).contiguous()
    n_runs = int(N.query("trec_group_pairs_lds_runs", int(n_pairs), int(n_items))) \
        if workspace_with_counts is None and ranks is None and N.load().trec_get_tuning(b"group_pairs_lds", 1) != 0 else 0
    if n_runs > 0:
        # few buckets, very many pairs (MovieLens-shaped catalogues): counters in LDS, no global atomics (csrc/segment.hip)
        ws32 = torch.empty((n_items,), dtype=torch.int32, device=dev)
        ws64 = torch.empty(((n_items + 1023) // 1024 + 1,), dtype=torch.int64, device=dev)
        run_counts = torch.empty((n_runs, n_items), dtype=torch.int32, device=dev)
        indptr_t = torch.empty((n_items + 1,), dtype=torch.int64, device=dev)
        users_t = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
        perm_t = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
        with _timed("group_pairs_by_item"):
            N.call("trec_group_pairs_by_item_lds", N.ptr(xu32), N.ptr(xi32), n_pairs, pairs_per_user, n_items, N.ptr(ws32),
                   N.ptr(ws64), N.ptr(run_counts), N.ptr(indptr_t), N.ptr(users_t), N.ptr(perm_t))
        return indptr_t, users_t, perm_t
    ws32 = workspace_with_counts if workspace_with_counts is not None else \
        torch.empty((2 * n_items,), dtype=torch.int32, device=dev)
    ws64 = torch.empty(((n_items + 1023) // 1024 + 1,), dtype=torch.int64, device=dev)
    indptr_t = torch.empty((n_items + 1,), dtype=torch.int64, device=dev)
    carry = values is not None and ranks is not None
    users_t = None if carry else torch.empty((n_pairs,), dtype=torch.int32, device=dev)
    if carry:      # packed: entries int2 {user, value bits}, one 8-byte scattered store per pair; (indptr, entries, None)
        entries = torch.empty((n_pairs, 2), dtype=torch.int32, device=dev)
        lib = N.load()
        if workspace_with_counts is not None and lib.trec_get_tuning(b"group_pairs_staged", 1) != 0 and \
                n_pairs >= lib.trec_get_tuning(b"group_pairs_staged_min", 1 << 24):
            # very many pairs: the fill in two levels (csrc/segment.hip) -- appended to the staging region of their destination
            # window first, placed window by window afterwards, instead of 8-byte stores scattered over the whole 800 MB of
            # entries.  Windows of 2^22 entries measured best at 1e8 pairs (fill + scan 3.1 ms against 3.7 in one level; 2^17:
            # 6.0, 2^19: 4.4, 2^21: 3.5, 2^23: 3.5, 2^24: 3.7 -- the staging pass pays per window, the placing pass per entry)
            wlog = int(lib.trec_get_tuning(b"group_pairs_window_log2", 22))
            sbytes = int(N.query("trec_group_pairs_staged_bytes", int(n_pairs), wlog))
            if sbytes > 0:
                staging = torch.empty((sbytes,), dtype=torch.uint8, device=dev)
                with _timed("group_pairs_staged"):
                    N.call("trec_group_pairs_by_item_staged", N.ptr(xu32), N.ptr(xi32), n_pairs, pairs_per_user, n_items,
                           N.ptr(ws32), N.ptr(ws64), N.ptr(indptr_t), N.ptr(entries), N.ptr(ranks), N.ptr(values),
                           N.ptr(staging), sbytes, wlog)
                return indptr_t, entries, None
        N.call("trec_group_pairs_by_item", N.ptr(xu32), N.ptr(xi32), n_pairs, pairs_per_user, n_items, N.ptr(ws32),
               N.ptr(ws64), N.ptr(indptr_t), N.ptr(entries), None, 1, N.ptr(ranks), N.ptr(values), None)
        return indptr_t, entries, None
    perm_t = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
    N.call("trec_group_pairs_by_item", N.ptr(xu32), N.ptr(xi32), n_pairs, pairs_per_user, n_items, N.ptr(ws32),
           N.ptr(ws64), N.ptr(indptr_t), N.ptr(users_t), N.ptr(perm_t), 1 if workspace_with_counts is not None else 0,
           N.ptr(ranks), None, None)
    return indptr_t, users_t, perm_t


_ones_cache = {}


def _ones(n, dev):
    key = (n, str(dev))
    if key not in _ones_cache:
        _ones_cache.clear()
        _ones_cache[key] = torch.ones((n,), dtype=torch.float32, device=dev)
    return _ones_cache[key]


def pair_score(user_repr, item_repr, x_user, x_item, mode=MODE_DOT, user_bias=None, item_bias=None):
    """Serial prediction for (x_user[p], x_item[p]) pairs, optionally fused with the serial bias add."""
    xu32, ppu = _idx32(x_user)
    xi32, _ = _idx32(x_item)
    inter = getattr(x_user, "interactions", None) if isinstance(x_user, PairIndex) else None
    if inter is not None and (getattr(x_item, "interactions", None) is not inter or xi32.numel() != inter.nnz):
        inter = None
    if ppu > 0:
        xu32 = None                      # implicit users: pair p -> p // pairs_per_user
    ub = _f32c(user_bias) if user_bias is not None else None
    ib = _f32c(item_bias) if item_bias is not None else None
    return _PairScore.apply(user_repr, item_repr, ub, ib, xu32, xi32, ppu, mode, inter)


# ------------------------------------------------------------------------------------------------ K9
def _pair_bias_grads(g, xu32, xi32, ppu, inter, n_users, n_items, want_ub, want_ib):
    """d user_bias / d item_bias of out[p] = ... + ub[user(p)] + ib[item(p)]: g summed per user / per item with the
    segmented K1 matvec over the pair structure (interactions: CSR + its transpose; samples: ppu consecutive pairs per
    user + a device counting sort by item).  Unstructured pair lists fall back to index_add_."""
    dev = g.device
    n_pairs = xi32.numel()
    dub = dib = None
    if inter is None and ppu <= 0:
        if want_ub:
            dub = torch.zeros((n_users,), dtype=torch.float32, device=dev).index_add_(0, xu32.long(), g)
        if want_ib:
            dib = torch.zeros((n_items,), dtype=torch.float32, device=dev).index_add_(0, xi32.long(), g)
        return dub, dib
    if want_ub:
        indptr_u = inter.indptr if inter is not None else \
            torch.arange(0, (n_users + 1) * ppu, ppu, dtype=torch.int64, device=dev)
        long_u = (inter.max_row_nnz if inter is not None else ppu) > SPLIT_T
        dub = spmv_raw(indptr_u, xi32, g, None, n_users, n_pairs, None, long_u)
    if want_ib:
        indptr_t, users_t, perm_t = inter.transposed() if inter is not None else \
            group_pairs_by_item(xu32, xi32, ppu, n_items)
        long_i = inter.max_col_nnz > SPLIT_T if inter is not None else _sampled_buckets_long(n_pairs, n_items)
        dib = spmv_raw(indptr_t, users_t, g, perm_t, n_items, n_pairs, None, long_i)
    return dub, dib


class _CollapseTastes(torch.autograd.Function):
    """K9 forward / backward; ``meta`` = (bias_mode, xu32, xi32, span, inter)."""

    @staticmethod
    def forward(ctx, preds, attn, ub, ib, meta):
        bias_mode, xu32, xi32, span, inter = meta
        preds = _f32c(preds)
        attn = _f32c(attn) if attn is not None else None
        T = preds.shape[0]
        n = preds[0].numel()
        out = torch.empty(preds.shape[1:], dtype=torch.float32, device=preds.device)
        N.call("trec_collapse_tastes_fwd", N.ptr(preds), N.ptr(attn), T, n, bias_mode, N.ptr(ub), N.ptr(ib),
               N.ptr(xu32), N.ptr(xi32), span, N.ptr(out))
        ctx.save_for_backward(preds, attn)
        ctx.meta = meta
        ctx.bias_shapes = (None if ub is None else ub.numel(), None if ib is None else ib.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        preds, attn = ctx.saved_tensors
        bias_mode, xu32, xi32, span, inter = ctx.meta
        n_ub, n_ib = ctx.bias_shapes
        g = _f32c(g)
        T = preds.shape[0]
        n = preds[0].numel()
        d_preds = torch.empty_like(preds)
        d_attn = torch.empty_like(attn) if attn is not None else None
        N.call("trec_collapse_tastes_bwd", N.ptr(preds), N.ptr(attn), N.ptr(g), T, n, N.ptr(d_preds), N.ptr(d_attn))
        dub = dib = None
        if bias_mode == 1 and (n_ub is not None or n_ib is not None):
            ppu = span if xu32 is None else 0
            n_users = inter.shape[0] if inter is not None else (n // ppu if ppu > 0 else (n_ub or 0))
            n_items = inter.shape[1] if inter is not None else (n_ib or 0)
            dub, dib = _pair_bias_grads(g.reshape(-1), xu32, xi32, ppu, inter, n_users, n_items, n_ub is not None,
                                        n_ib is not None)
        elif bias_mode == 2:
            g2 = g.reshape(-1, span)
            if n_ub is not None:
                dub = gemm_raw(g2, _ones(span, g.device).reshape(span, 1)).reshape(-1)
            if n_ib is not None:
                dib = colsum(g2)
        return d_preds, d_attn, dub, dib, None


def collapse_tastes(tastes_predictions, tastes_attentions=None, user_bias=None, item_bias=None, x_user=None,
                    x_item=None):
    """collapse_mixture_of_tastes (recommendation_graphs.py:85-109) in one pass (K9), optionally fused with the bias add
    that follows it: serial predictions pass the pair indices, dense [n_users, n_items] predictions pass none."""
    def stacked(x):
        return x if isinstance(x, torch.Tensor) else torch.stack(list(x))

    preds = stacked(tastes_predictions)
    attn = stacked(tastes_attentions) if tastes_attentions is not None else None
    if preds.shape[0] > 16:
        # the collapse kernel holds <= 16 tastes: a maximum is a maximum of group maxima (exact, bias add on the last
        # call only); the attention form (softmax over ALL tastes) is composed from torch ops -- same arithmetic order as
        # recommendation_graphs.py:96-105 (stack, softmax, multiply, reduce_sum), then the bias adds
        if attn is None:
            groups = [collapse_tastes(preds[g:g + 16]) if preds[g:g + 16].shape[0] > 1 else preds[g]
                      for g in range(0, preds.shape[0], 16)]
            return collapse_tastes(torch.stack(groups), None, user_bias, item_bias, x_user, x_item)
        out = (preds * torch.softmax(attn, dim=0)).sum(dim=0)
        if user_bias is None and item_bias is None:
            return out
        if x_item is not None:
            xu, xi = x_user.long(), x_item.long()
            if user_bias is not None:
                out = out + user_bias[xu]
            if item_bias is not None:
                out = out + item_bias[xi]
            return out
        if user_bias is not None:
            out = out + user_bias.reshape(-1, 1)
        if item_bias is not None:
            out = out + item_bias.reshape(1, -1)
        return out
    ub = _f32c(user_bias) if user_bias is not None else None
    ib = _f32c(item_bias) if item_bias is not None else None
    if ub is None and ib is None:
        meta = (0, None, None, 0, None)
    elif x_item is not None:
        xu32, ppu = _idx32(x_user)
        xi32, _ = _idx32(x_item)
        inter = getattr(x_user, "interactions", None) if isinstance(x_user, PairIndex) else None
        if inter is not None and (getattr(x_item, "interactions", None) is not inter or xi32.numel() != inter.nnz):
            inter = None
        if ppu > 0:
            xu32 = None
        meta = (1, xu32, xi32, ppu, inter)
    else:
        if preds.dim() != 3:
            raise ValueError("dense bias add needs [n_tastes, n_users, n_items] predictions")
        meta = (2, None, None, preds.shape[2], None)
    return _CollapseTastes.apply(preds, attn, ub, ib, meta)


# ------------------------------------------------------------------------------------------------ K6
class _WMRB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_serial, sample_pred, inter: Interactions, weight):
        pred_serial, sample_pred = _f32c(pred_serial), _f32c(sample_pred)
        n_users, n_items = inter.shape
        S = sample_pred.shape[1]
        loss = torch.empty((inter.n_positive,), dtype=torch.float32, device=pred_serial.device)
        smr = torch.empty_like(loss)
        N.call("trec_wmrb_fwd", N.ptr(inter.indptr), N.ptr(inter.pos_slot), N.ptr(weight), N.ptr(pred_serial),
               N.ptr(sample_pred), n_users, n_items, S, N.ptr(loss), N.ptr(smr))
        ctx.inter, ctx.weight = inter, weight
        ctx.save_for_backward(pred_serial, sample_pred, smr)
        return loss

    @staticmethod
    def backward(ctx, gl):
        pred_serial, sample_pred, smr = ctx.saved_tensors
        inter = ctx.inter
        n_users, n_items = inter.shape
        S = sample_pred.shape[1]
        dp = torch.empty_like(pred_serial)
        ds = torch.empty_like(sample_pred)
        N.call("trec_wmrb_bwd", N.ptr(inter.indptr), N.ptr(inter.pos_slot), N.ptr(ctx.weight), N.ptr(pred_serial),
               N.ptr(sample_pred), N.ptr(smr), N.ptr(_f32c(gl)), n_users, n_items, S, N.ptr(dp), N.ptr(ds))
        return dp, ds, None, None


import os as _os

# sampled pairs from which the rank-free binned grouping replaces histogram atomics + ranked fill (TREC_BINNED_MIN_PAIRS: A/B runs)
LAST_FUSED_STATS = {}          # filled while KERNEL_EVENTS is collecting: sampled pairs / the pairs the item side kept (device scalar)
GROUP_BINNED_MIN_PAIRS = int(_os.environ.get("TREC_BINNED_MIN_PAIRS", 1 << 22))     # (2^22: the kernels' own minimum; one rank of 8 at
                                                                                     # 1M x 1M has 1.25e7 pairs: 3.91 vs 4.10 ms per step)


def wmrb_fused_supported(n_sampled, interactions, d):
    """Can the one-pass WMRB step (csrc/wmrb_fused.hip) run this shape?  (LDS holds S + max interactions-per-user rows.)"""
    if not N.load().trec_get_tuning(b"wmrb_fused", 1):
        return False
    return N.query("trec_wmrb_fused_lds_bytes", int(n_sampled), int(interactions.max_row_nnz), int(d)) >= 0


def wmrb_fused_step(user_in, item_in, user_bias, item_bias, interactions, samples, balanced=False):
    """One WMRB / BalancedWMRB step for dot scores on (user_in, item_in), upstream gradient 1 (the trainer minimises
    the SUM of the loss vector): returns (loss [P+], pred_serial [P], d user_in, d item_in, d user_bias, d item_bias).
    The user side -- predictions of the interactions and of the sampled pairs, loss, dU, d b_u -- is one kernel with
    every item row gathered once; the item side groups the per-pair coefficients by item (static transposed structure
    for the interactions, device counting sort for the samples) and runs the segmented K1 gathers / segment sums."""
    u, v = _f32c(user_in.detach()), _f32c(item_in.detach())
    ub = _f32c(user_bias.detach()) if user_bias is not None else None
    ib = _f32c(item_bias.detach()) if item_bias is not None else None
    n_users, n_items = interactions.shape
    S = int(samples.shape[1])
    d = u.shape[1]
    dev = u.device
    nnz = interactions.nnz
    loss = torch.empty((interactions.n_positive,), dtype=torch.float32, device=dev)
    pred = torch.empty((nnz,), dtype=torch.float32, device=dev)
    d_u = torch.empty_like(u)
    d_ub = torch.empty((n_users,), dtype=torch.float32, device=dev) if ub is not None else None
    coef_s = torch.empty((n_users, S), dtype=torch.float32, device=dev)
    coef_p = torch.empty((nnz,), dtype=torch.float32, device=dev)
    weight = interactions.balanced_weight() if balanced else None
    samples = samples.to(torch.int32).contiguous()
    xs = samples.reshape(-1)
    # very many sampled pairs: grouped by item through a two-level LDS partition that needs no ranks (csrc/segment.hip, "binned") --
    # the fused kernel then issues no histogram atomics at all; otherwise the histogram rides in the fused kernel and the ranks its
    # atomics return make the fill atomic-free
    binned_bytes = 0
    if xs.numel() >= GROUP_BINNED_MIN_PAIRS and not _LOCAL.deterministic_grouping and \
            N.load().trec_get_tuning(b"group_pairs_binned", 1) != 0:
        binned_bytes = int(N.query("trec_group_pairs_binned_bytes", int(xs.numel()), int(n_items)))
    if binned_bytes > 0:
        ws32 = ranks = None
    else:
        ws32 = torch.zeros((2 * n_items,), dtype=torch.int32, device=dev)     # [sample histogram | cursors] of the sort below
        ranks = torch.empty((n_users, S), dtype=torch.int32, device=dev)
    with _timed("wmrb_fused_step"):
        N.call("trec_wmrb_fused_step", N.ptr(u), N.ptr(v), N.ptr(ub), N.ptr(ib), N.ptr(interactions.indptr),
               N.ptr(interactions.x_item32), N.ptr(interactions.pos_slot), N.ptr(weight), N.ptr(samples), n_users,
               n_items, S, d, int(interactions.max_row_nnz), N.ptr(loss), N.ptr(pred), N.ptr(d_u), N.ptr(d_ub),
               N.ptr(coef_s), N.ptr(coef_p), N.ptr(ws32), N.ptr(ranks))
    # ---- item side: d item_in = G^T . user_in over both pair lists, d b_i = per-item sums of the coefficients
    # (MEASURED and dropped, profiles/r05_fit_overlap_ab.txt: the sort on a second stream -- whole, next to the fused kernel with
    # the coefficients read through the pair permutation: 34.4 ms per epoch against 25.7; only its fill, next to the item-side
    # gather of the interactions: 26.2 against 26.1 -- every kernel of this step is bound by the same fabric, an overlapped pair
    # slows down by exactly what it hides)
    d_v = torch.zeros_like(v) if nnz == 0 else None
    d_ib = torch.zeros((n_items,), dtype=torch.float32, device=dev) if ib is not None else None
    # (epilogue 3 of K1: the row sums of the gathered coefficients = d b_i come out of the same pass)
    epi = EPI_ROWSUM if ib is not None else EPI_NONE
    rowsum = d_ib if epi == EPI_ROWSUM else None
    if nnz:
        indptr_t, users_t, perm_t = interactions.transposed()
        if interactions.max_col_nnz > SPLIT_T:           # popular items: their pairs are summed as chunks
            d_v = spmm_split(indptr_t, users_t, coef_p, perm_t, n_items, nnz, u, want_rowsum=rowsum)
        else:
            d_v = _spmm_rowsum(indptr_t, users_t, coef_p, perm_t, n_items, nnz, u, epi, False, None, d_ib)
    split_samples = xs.numel() > n_items * _SPLIT_MEAN
    if binned_bytes > 0:
        ind_s = torch.empty((n_items + 1,), dtype=torch.int64, device=dev)
        entries = torch.empty((xs.numel(), 2), dtype=torch.int32, device=dev)
        bws = torch.empty((binned_bytes,), dtype=torch.uint8, device=dev)
        # a sample that violates no margin of its user has coefficient exactly 0 and adds nothing to its item's gradient: such pairs
        # are left out of the sort and of the item-side gather (the chunked gather below is sized by the pair count: all pairs there)
        drop_zero = 0 if split_samples else int(N.load().trec_get_tuning(b"group_pairs_drop_zero", 1) != 0)
        with _timed("group_pairs_binned"):
            N.call("trec_group_pairs_by_item_binned", None, N.ptr(xs), N.ptr(coef_s.reshape(-1)), int(xs.numel()), S, n_items,
                   drop_zero, N.ptr(bws), binned_bytes, N.ptr(ind_s), N.ptr(entries))
    else:
        ind_s, entries, _ = group_pairs_by_item(None, xs, S, n_items, workspace_with_counts=ws32, ranks=ranks.reshape(-1),
                                                values=coef_s.reshape(-1))
    # which grouping this call took (host-side facts only: parity records and tests assert the route they mean to cover)
    LAST_FUSED_STATS["route"] = "binned" if binned_bytes > 0 else "ranked"
    LAST_FUSED_STATS["drop_zero"] = bool(binned_bytes > 0 and drop_zero)
    if KERNEL_EVENTS is not None:                        # (measurement runs only: how many sampled pairs the item side still sees)
        LAST_FUSED_STATS["sampled_pairs"] = int(xs.numel())
        LAST_FUSED_STATS["sampled_pairs_kept"] = ind_s[-1:].clone()
        LAST_FUSED_STATS["sampled_pairs_with_coefficient_0"] = (coef_s == 0).sum()
        LAST_FUSED_STATS["loss_mean"] = loss.mean()
    if split_samples:                                    # few items: every bucket of samples is long
        spmm_split(ind_s, None, None, None, n_items, xs.numel(), u, accumulate=True, out=d_v, want_rowsum=rowsum,
                   packed=entries)
    else:
        with _timed("spmm_csr"):
            N.call("trec_spmm_csr_packed", N.ptr(ind_s), N.ptr(entries), n_items, N.ptr(u), d, epi, 1, N.ptr(d_v),
                   N.ptr(rowsum))
    return loss, pred, d_u, d_v, d_ub, d_ib


DENSE_G_MIN_DENSITY = 1.0 / 32.0      # pairs per user / items from which the item side of the tiled step is a GEMM (G^T . U)


def wmrb_tiled_supported(n_sampled, interactions, d):
    """Can the tiled one-pass WMRB step (csrc/wmrb_tiled.hip: any S, dot or Euclidean scores) run this shape?"""
    if not N.load().trec_get_tuning(b"wmrb_tiled", 1):
        return False
    return N.query("trec_wmrb_tiled_lds_bytes", int(n_sampled), int(interactions.max_row_nnz), int(d)) >= 0


def _dense_g_fits(n_users, n_items, n_sampled, nnz, dev):
    """The dense coefficient matrix G [n_users, n_items] pays when the pairs of a user are a sizeable share of the items (the
    GEMM does n_items * d MACs per user on fp32 MFMA, the gathers move pairs * d * 4 bytes twice) and fits beside everything else."""
    if not N.load().trec_get_tuning(b"wmrb_dense_g", 1) or n_users == 0:
        return False
    if (n_sampled + nnz / float(n_users)) < DENSE_G_MIN_DENSITY * n_items:
        return False
    need = float(n_users) * ((n_items + 3) // 4 * 4) * 4.0
    if need <= float(1 << 30):                # (small models: no memory query -- the step may be inside a HIP-graph capture)
        return True
    if torch.cuda.is_current_stream_capturing():
        return False
    free, _ = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    return need <= 0.4 * free


def wmrb_tiled_step(user_in, item_in, user_bias, item_bias, interactions, samples, balanced=False, mode=MODE_DOT):
    """wmrb_fused_step for S in the thousands, long rows and Euclidean scores: (loss [P+], pred_serial [P], d user_in, d item_in,
    d user_bias, d item_bias), upstream gradient 1.  User side: ONE kernel, the user's item rows streamed twice (scores into LDS,
    then dU).  Item side: when a user's pairs are a sizeable share of the catalogue (configs[4]: 10 %), the pairs' values were
    also added into a dense G [n_users, n_items] and d item_in is G^T . U on fp32 MFMA -- no sort, no second gather of 3.7e8
    rows; otherwise the grouped gathers of the composed path (transposed interactions + counting sort of the samples)."""
    u, v = _f32c(user_in.detach()), _f32c(item_in.detach())
    ub = _f32c(user_bias.detach()) if user_bias is not None else None
    ib = _f32c(item_bias.detach()) if item_bias is not None else None
    n_users, n_items = interactions.shape
    S = int(samples.shape[1])
    d = u.shape[1]
    dev = u.device
    nnz = interactions.nnz
    euclid = mode == MODE_EUCLIDEAN
    loss = torch.empty((interactions.n_positive,), dtype=torch.float32, device=dev)
    pred = torch.empty((nnz,), dtype=torch.float32, device=dev)
    d_ub = torch.empty((n_users,), dtype=torch.float32, device=dev) if ub is not None else None
    val_s = torch.empty((n_users, S), dtype=torch.float32, device=dev)
    val_p = torch.empty((nnz,), dtype=torch.float32, device=dev)
    need_raw = euclid and ib is not None
    raw_s = torch.empty((n_users, S), dtype=torch.float32, device=dev) if need_raw else None
    raw_p = torch.empty((nnz,), dtype=torch.float32, device=dev) if need_raw else None
    weight = interactions.balanced_weight() if balanced else None
    samples = samples.to(torch.int32).contiguous()
    xs = samples.reshape(-1)
    dense = _dense_g_fits(n_users, n_items, S, nnz, dev) and not _LOCAL.deterministic_grouping
    ldg = (n_items + 3) // 4 * 4
    G = torch.zeros((n_users, ldg), dtype=torch.float32, device=dev) if dense else None
    # with G the user side needs no second sweep over the rows either: dU = G . V (euclidean: rowsum(G) U - G . V)
    d_u = None if dense else torch.empty_like(u)
    val_rs = torch.empty((n_users,), dtype=torch.float32, device=dev) if dense else None
    with _timed("wmrb_tiled_step"):
        N.call("trec_wmrb_tiled_step", N.ptr(u), N.ptr(v), N.ptr(ub), N.ptr(ib), N.ptr(interactions.indptr),
               N.ptr(interactions.x_item32), N.ptr(interactions.pos_slot), N.ptr(weight), N.ptr(samples), n_users, n_items, S, d,
               int(mode), int(interactions.max_row_nnz), N.ptr(loss), N.ptr(pred), N.ptr(d_u), N.ptr(d_ub), N.ptr(val_s),
               N.ptr(val_p), N.ptr(raw_s), N.ptr(raw_p), N.ptr(G), ldg, N.ptr(val_rs))
    LAST_FUSED_STATS["route"] = "tiled+dense_g" if dense else "tiled+grouped"
    LAST_FUSED_STATS["drop_zero"] = False
    d_ib = None
    if dense:
        # d item_in: dot  dV[i] = sum_u G[u, i] U[u];  euclidean  dV[i] = sum c (V[i] - U[u]) = colsum(G)[i] V[i] - (G^T U)[i]
        v_pad = v if ldg == n_items else torch.cat([v, torch.zeros((ldg - n_items, d), dtype=torch.float32, device=dev)])
        # (both GEMMs are gradients, held to 1e-4: split-bf16 operands on bf16 MFMA, bound by reading G instead of by the fp32
        # matrix pipe -- tuning dense_g_split_bf16 = 0: exact fp32 products)
        split = N.load().trec_get_tuning(b"dense_g_split_bf16", 1) != 0
        with _timed("dense_g_gemm"):
            gv = gemm_raw(G, v_pad, split_bf16=split)          # [n_users, d]
        d_u = val_rs.unsqueeze(1) * u - gv if euclid else gv
        want_cs = euclid or ib is not None
        with _timed("dense_g_gemm"):                          # (the column sums of G ride in the threads that stage it)
            t = gemm_raw(G, u, trans_a=True, split_bf16=split, a_colsum=want_cs and split)
            t, cs = t if (want_cs and split) else (t, None)
            t = t[:n_items]
        if want_cs:
            cs = cs[:n_items] if cs is not None else colsum(G)[:n_items]
        d_v = cs.unsqueeze(1) * v - t if euclid else t.contiguous()
        if ib is not None and not euclid:
            d_ib = cs.contiguous()
        del G
    else:
        d_v = torch.zeros_like(v) if nnz == 0 else None
        want_rs = ib is not None and not euclid
        if want_rs:
            d_ib = torch.zeros((n_items,), dtype=torch.float32, device=dev)
        if nnz:
            indptr_t, users_t, perm_t = interactions.transposed()
            d_v = spmm_split(indptr_t, users_t, val_p, perm_t, n_items, nnz, u, own=v if euclid else None,
                             want_rowsum=d_ib if want_rs else False)
        ind_s, users_s, perm_s = group_pairs_by_item(None, xs, S, n_items)
        spmm_split(ind_s, users_s, val_s.reshape(-1), perm_s, n_items, xs.numel(), u, own=v if euclid else None,
                   accumulate=True, out=d_v, want_rowsum=d_ib if want_rs else False)      # (row sums accumulate with the rows)
    if need_raw:
        d_ib = torch.zeros((n_items,), dtype=torch.float32, device=dev)
        with _timed("item_weighted_hist"):
            N.call("trec_item_weighted_hist", N.ptr(xs), N.ptr(raw_s.reshape(-1)), int(xs.numel()), N.ptr(interactions.x_item32),
                   N.ptr(raw_p), int(nnz), int(n_items), N.ptr(d_ib))
    return loss, pred, d_u, d_v, d_ub, d_ib


def _spmm_rowsum(indptr, indices, values, perm, n_rows, nnz, w, epilogue, accumulate, out, rowsum):
    w = _f32c(w)
    if out is None:
        out = torch.empty((n_rows, w.shape[1]), dtype=torch.float32, device=w.device)
    with _timed("spmm_csr"):
        N.call("trec_spmm_csr", N.ptr(indptr), N.ptr(indices), N.ptr(values), N.ptr(perm), n_rows, nnz, N.ptr(w),
               w.shape[1], None, epilogue, 1 if accumulate else 0, N.ptr(out), N.ptr(rowsum) if epilogue == EPI_ROWSUM else None)
    return out


def wmrb_loss(pred_serial, sample_pred, interactions, balanced=False):
    if sample_pred.shape[0] != interactions.shape[0]:
        raise ValueError("tf_sample_predictions must have one row per user")
    weight = interactions.balanced_weight() if balanced else None
    return _WMRB.apply(pred_serial, sample_pred, interactions, weight)


class _RMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y):
        pred, y = _f32c(pred), _f32c(y)
        n = pred.numel()
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        n_div = n
        if n:
            n_partial = max(1, min(1024, (n + 4095) // 4096))
            ws = torch.empty((n_partial,), dtype=torch.float32, device=pred.device)
            N.call("trec_rmse_fwd", N.ptr(y), N.ptr(pred), n, N.ptr(ws), n_partial, N.ptr(loss))
        else:
            loss.zero_()
        if _LOCAL.loss_group is not None:
            # user shards: sqrt(sum of every rank's squared errors / all interactions) -- this rank's sum is loss^2 * n; the
            # backward kernel divides by (n_div * loss), now the global count and the global loss
            sums = torch.stack([loss.double()[0] ** 2 * float(n), torch.tensor(float(n), dtype=torch.float64, device=pred.device)])
            _loss_all_reduce(sums)
            loss = torch.sqrt(sums[0] / sums[1]).to(torch.float32).reshape(1)
            n_div = int(sums[1].item())
        ctx.save_for_backward(pred, y, loss)
        ctx.n_div = n_div
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        pred, y, loss = ctx.saved_tensors
        dp = torch.empty_like(pred)
        n = pred.numel()
        g = _f32c(gl).reshape(1)
        if ctx.n_div != n and n:
            g = g * (float(n) / float(ctx.n_div))         # (the kernel's divisor is its own n)
        if n:
            N.call("trec_rmse_bwd", N.ptr(y), N.ptr(pred), N.ptr(loss), N.ptr(g), n, N.ptr(dp))
        return dp, None


def rmse_loss(pred_serial, interactions_serial):
    return _RMSE.apply(pred_serial, interactions_serial)


# ------------------------------------------------------------------------------------------------ dense / separation losses
DENSE_LOSS_SEPARATION, DENSE_LOSS_SEPARATION_DENSE, DENSE_LOSS_RMSE_DENSE = 0, 1, 2


class _DenseLoss(torch.autograd.Function):
    """RMSEDense / Separation / SeparationDense (loss_graphs.py:62-134) as streaming reductions (csrc/loss_dense.hip): the
    predictions are read once per pass, the interactions enter as their sparse list, the statistics stay on the device in a
    double[16] block that the backward kernels read."""

    @staticmethod
    def forward(ctx, pred, kind, xu32, xi32, values):
        pred = _f32c(pred)
        if kind == DENSE_LOSS_SEPARATION:
            rows, cols = int(pred.numel()), 1
        else:
            rows, cols = int(pred.shape[0]), int(pred.shape[1])
        n_pairs = int(values.numel())
        st = torch.empty((16,), dtype=torch.float64, device=pred.device)
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        if _LOCAL.loss_group is None:
            N.call("trec_dense_loss_fwd", kind, N.ptr(pred), rows, cols, N.ptr(xu32), N.ptr(xi32), N.ptr(values), n_pairs, N.ptr(st),
                   N.ptr(loss))
        else:
            # user shards: the sums of both passes are those of every rank's predictions (csrc/loss_dense.hip, "phase")
            n_all = torch.tensor([float(rows * cols)], dtype=torch.float64, device=pred.device)
            n_all_total = int(_loss_all_reduce(n_all).item()) if kind != DENSE_LOSS_SEPARATION else 0
            for phase in (0, 1, 2):
                N.call("trec_dense_loss_fwd_phase", kind, phase, N.ptr(pred), rows, cols, N.ptr(xu32), N.ptr(xi32), N.ptr(values),
                       n_pairs, n_all_total, N.ptr(st), N.ptr(loss))
                if phase == 0 or (phase == 1 and kind != DENSE_LOSS_RMSE_DENSE):
                    _loss_all_reduce(st[:10])
        ctx.save_for_backward(pred, st)
        ctx.meta = (kind, rows, cols, xu32, xi32, values, n_pairs)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        pred, st = ctx.saved_tensors
        kind, rows, cols, xu32, xi32, values, n_pairs = ctx.meta
        d_pred = torch.empty_like(pred)
        N.call("trec_dense_loss_bwd", kind, N.ptr(pred), rows, cols, N.ptr(xu32), N.ptr(xi32), N.ptr(values), n_pairs, N.ptr(st),
               N.ptr(_f32c(gl).reshape(1)), N.ptr(d_pred))
        return d_pred, None, None, None, None


class FactoredPrediction(object):
    """The dense prediction of a dot-product (or cosine: rows normalised) model kept as its FACTORS -- user / item
    representations, biases -- plus the serial predictions of the interactions.  TensorRec hands it to the built-in dense losses
    in place of the [n_users, n_items] tensor (which is 4 TB at 1M x 1M): they need two Gram matrices, not the predictions."""

    def __init__(self, user_repr, item_repr, user_bias, item_bias, pred_serial):
        self.user_repr, self.item_repr, self.user_bias, self.item_bias, self.pred_serial = \
            user_repr, item_repr, user_bias, item_bias, pred_serial
        self.shape = (int(user_repr.shape[0]), int(item_repr.shape[0]))


def _augment(rows, bias, ones_first):
    """X = [u | b_u | 1] (ones_first False) / Y = [v | 1 | b_i] (True), float32, contiguous; a missing bias is a zero column"""
    n = rows.shape[0]
    one = torch.ones((n, 1), dtype=torch.float32, device=rows.device)
    b = bias.detach().reshape(n, 1).to(torch.float32) if bias is not None else torch.zeros_like(one)
    return torch.cat([rows.detach().to(torch.float32), one, b] if ones_first else [rows.detach().to(torch.float32), b, one], dim=1).contiguous()


class _FactoredDenseLoss(torch.autograd.Function):
    """RMSEDense / SeparationDense (loss_graphs.py:62-72, :100-134) of p_ui = x_u . y_i from the Gram matrices X^T X, Y^T Y (double,
    csrc/loss_dense.hip): forward O((U + I) d^2) instead of O(U I d), no [U, I] tensor; backward dX = A X (Y^T Y) + B 1 (sum y)^T
    on fp32 MFMA.  The interaction cells' corrections flow into the serial predictions' gradient (K3 backward)."""

    @staticmethod
    def forward(ctx, user_repr, item_repr, user_bias, item_bias, pred_serial, kind, values):
        X, Y = _augment(user_repr, user_bias, False), _augment(item_repr, item_bias, True)
        dev, D, d = X.device, X.shape[1], user_repr.shape[1]
        gx = torch.empty((D, D), dtype=torch.float64, device=dev)
        gy = torch.empty((D, D), dtype=torch.float64, device=dev)
        with _timed("gram_f64"):
            N.call("trec_gram_f64", N.ptr(X), X.shape[0], D, D, N.ptr(gx))
            N.call("trec_gram_f64", N.ptr(Y), Y.shape[0], D, D, N.ptr(gy))
        # sum p = (sum x) . (sum y): the column of ones makes row d + 1 of X^T X the column sums of X (row d of Y^T Y those of Y)
        m = torch.stack([(gx[d + 1] * gy[d]).sum(), (gx * gy).sum()]).contiguous()
        ps = _f32c(pred_serial.detach()).reshape(-1)
        n_pairs = int(values.numel())
        n_local = int(X.shape[0]) * int(Y.shape[0])
        st = torch.empty((16,), dtype=torch.float64, device=dev)
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        n_total = n_local
        if _LOCAL.loss_group is not None:
            n_total = int(_loss_all_reduce(torch.tensor([float(n_local)], dtype=torch.float64, device=dev)).item())
        for phase in (0, 1, 2):
            N.call("trec_dense_loss_factored_phase", kind, phase, N.ptr(m), N.ptr(ps), N.ptr(values), n_pairs, n_local, n_total,
                   N.ptr(st), N.ptr(loss))
            if _LOCAL.loss_group is not None and (phase == 0 or (phase == 1 and kind != DENSE_LOSS_RMSE_DENSE)):
                _loss_all_reduce(st[:10])
        ctx.save_for_backward(X, Y, gx, gy, ps, st)
        ctx.meta = (kind, values, n_pairs, d, user_bias is not None, item_bias is not None)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        X, Y, gx, gy, ps, st = ctx.saved_tensors
        kind, values, n_pairs, d, has_ub, has_ib = ctx.meta
        dev = X.device
        d_serial = torch.empty_like(ps)
        coef = torch.empty((2,), dtype=torch.float64, device=dev)
        N.call("trec_dense_loss_factored_bwd", kind, N.ptr(ps), N.ptr(values), n_pairs, N.ptr(st), N.ptr(_f32c(gl).reshape(1)),
               N.ptr(d_serial), N.ptr(coef))
        a, b = coef[0].to(torch.float32), coef[1].to(torch.float32)
        # dX = A X Gy + B 1 sy^T, dY = A Y Gx + B 1 sx^T   (sx = row d + 1 of Gx, sy = row d of Gy); the scalars stay on the device
        dX = gemm_raw(X, gy.to(torch.float32)) * a + (gy[d].to(torch.float32) * b).unsqueeze(0)
        dY = gemm_raw(Y, gx.to(torch.float32)) * a + (gx[d + 1].to(torch.float32) * b).unsqueeze(0)
        du, dv = dX[:, :d].contiguous(), dY[:, :d].contiguous()
        dub = dX[:, d].contiguous() if has_ub else None            # X = [u | b_u | 1]
        dib = dY[:, d + 1].contiguous() if has_ib else None        # Y = [v | 1 | b_i]
        return du, dv, dub, dib, d_serial, None, None


def factored_dense_loss(pred, interactions, kind):
    ub = pred.user_bias.reshape(-1) if pred.user_bias is not None else None
    ib = pred.item_bias.reshape(-1) if pred.item_bias is not None else None
    return _FactoredDenseLoss.apply(pred.user_repr, pred.item_repr, ub, ib, pred.pred_serial, kind, interactions.values)


def separation_loss(pred_serial, interactions_serial):
    """SeparationLossGraph (loss_graphs.py:75-97): 1 - Normal(mu_n - mu_p, sqrt(var_n + var_p)).cdf(0) over the interactions"""
    return _DenseLoss.apply(pred_serial.reshape(-1), DENSE_LOSS_SEPARATION, None, None, _f32c(interactions_serial).reshape(-1))


def separation_dense_loss(prediction, interactions):
    """SeparationDenseLossGraph (loss_graphs.py:100-134): the same over every user-item pair, non-positives as negatives"""
    if isinstance(prediction, FactoredPrediction):
        return factored_dense_loss(prediction, interactions, DENSE_LOSS_SEPARATION_DENSE)
    return _DenseLoss.apply(prediction, DENSE_LOSS_SEPARATION_DENSE, interactions.x_user32, interactions.x_item32, interactions.values)


def rmse_dense_loss(prediction, interactions):
    """RMSEDenseLossGraph (loss_graphs.py:62-72): RMSE against the dense interaction matrix"""
    if isinstance(prediction, FactoredPrediction):
        return factored_dense_loss(prediction, interactions, DENSE_LOSS_RMSE_DENSE)
    return _DenseLoss.apply(prediction, DENSE_LOSS_RMSE_DENSE, interactions.x_user32, interactions.x_item32, interactions.values)



RANK_SORT_MAX = 32768            # csrc/rank.hip: the longest row one workgroup sorts in LDS


def rank_rows(scores):
    """rank_predictions (recommendation_graphs.py:73-82) of a score slab [n_users, n_items] -> int32 ranks, 1 = best,
    ties to the lower index.  Rows up to 32768 items are sorted in LDS by one workgroup; longer rows are sorted in
    32768-item chunks and every item binary-searches the other chunks (trec_rank_rows_chunked)."""
    scores = _f32c(scores)
    n_u, n_i = scores.shape
    ranks = torch.empty(scores.shape, dtype=torch.int32, device=scores.device)
    if n_i > RANK_SORT_MAX and n_u > 0 and N.load().trec_get_tuning(b"rank_chunked", 1):
        per_user = N.query("trec_rank_rows_workspace_bytes", 1, n_i)
        slab = int(max(1, min(65535, (4 << 30) // per_user, n_u)))
        nbytes = N.query("trec_rank_rows_workspace_bytes", slab, n_i)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=scores.device)
        for s in range(0, n_u, slab):
            e = min(s + slab, n_u)
            with _timed("rank_rows_chunked"):
                N.call("trec_rank_rows_chunked", N.ptr(scores[s:e]), e - s, n_i, scores.stride(0), N.ptr(ranks[s:e]),
                       ranks.stride(0), N.ptr(ws), nbytes)
        return ranks
    N.call("trec_rank_rows", N.ptr(scores), n_u, n_i, scores.stride(0), N.ptr(ranks), ranks.stride(0))
    return ranks


def pair_scores_exact(users_f32, items_f32, kpad, d, xu32, xi32, user_bias=None, item_bias=None, mode=MODE_DOT,
                      user_sq=None, item_sq=None, item_index_base=0):
    """Exact fp32 scores of (user row, GLOBAL item id) pairs: the k-ordered fmaf chain of the fp32 MFMA kernels and of the
    oracle, biases / Euclidean transform in the reference's order (trec_pair_score_exact)."""
    n_pairs = int(xi32.numel())
    out = torch.empty((n_pairs,), dtype=torch.float32, device=users_f32.device)
    N.call("trec_pair_score_exact", N.ptr(users_f32), N.ptr(items_f32), kpad, int(d), N.ptr(xu32), N.ptr(xi32), n_pairs,
           N.ptr(user_bias), N.ptr(item_bias), mode, N.ptr(user_sq), N.ptr(item_sq), int(item_index_base), N.ptr(out))
    return out


def rank_counts_fused(users_f32, items_f32, kpad, d, pair_indptr, xi32, user_bias=None, item_bias=None, mode=MODE_DOT,
                      user_sq=None, item_sq=None, item_index_base=0, n_chunks=0, target_scores=None):
    """Partial rank counts of pairs grouped by user, with NO score slab (csrc/score_rank.hip): the exact fp32 MFMA score
    tile is compared with the users' targets in registers.  ``users_f32`` / ``items_f32``: fp32 operands [n, kpad] from
    score_prep; ``pair_indptr``: host int64 [n_users + 1]; ``xi32``: device int32 [n_pairs] GLOBAL item ids of the pairs,
    sorted by user.  Returns int32 [n_pairs] counts over this call's items (rank = count + 1 once every item shard's
    counts are summed).  Users with more than 32 targets occupy several resident rows.  ``target_scores``: the pairs'
    exact scores when this call's items are a shard that does not hold every target (the owning shard computes them
    with pair_scores_exact); default: computed here from the operands."""
    import numpy as np
    dev = users_f32.device
    n_pairs = int(xi32.numel())
    counts = torch.zeros((n_pairs,), dtype=torch.int32, device=dev)
    if n_pairs == 0:
        return counts
    qmax = N.query("trec_score_rankcount_max_targets")
    indptr = np.asarray(pair_indptr, dtype=np.int64)
    per_user = np.diff(indptr)
    rows_per_user = -(-per_user // qmax)
    row_user = np.repeat(np.arange(len(per_user), dtype=np.int64), rows_per_user)
    first_row = np.concatenate([[0], np.cumsum(rows_per_user)[:-1]])
    j = np.arange(len(row_user), dtype=np.int64) - first_row[row_user]            # group number inside the user
    row_t0 = indptr[row_user] + qmax * j
    row_tn = np.minimum(qmax, per_user[row_user] - qmax * j)
    xu32 = torch.from_numpy(np.repeat(np.arange(len(per_user), dtype=np.int32), per_user)).to(dev)
    row_user_d = torch.from_numpy(row_user.astype(np.int32)).to(dev)
    row_t0_d = torch.from_numpy(row_t0.astype(np.int32)).to(dev)
    row_tn_d = torch.from_numpy(row_tn.astype(np.int32)).to(dev)
    tgt = target_scores if target_scores is not None else \
        pair_scores_exact(users_f32, items_f32, kpad, d, xu32, xi32, user_bias, item_bias, mode, user_sq, item_sq,
                          item_index_base)
    with _timed("score_gemm_rankcount"):
        N.call("trec_score_gemm_rankcount", N.ptr(users_f32), N.ptr(items_f32), kpad, len(row_user), items_f32.shape[0],
               int(item_index_base), N.ptr(user_bias), N.ptr(item_bias), mode, N.ptr(user_sq), N.ptr(item_sq),
               N.ptr(row_user_d), N.ptr(row_t0_d), N.ptr(row_tn_d), N.ptr(xi32), N.ptr(tgt), int(n_chunks), N.ptr(counts))
    return counts


def rank_of_pairs(scores, col_offset, begin, end, xu32, xi32, target_scores, add_one=True):
    out = torch.empty((xu32.numel(),), dtype=torch.int32, device=scores.device)
    N.call("trec_rank_of_pairs", N.ptr(scores), scores.stride(0), col_offset, begin, end, N.ptr(xu32), N.ptr(xi32),
           N.ptr(target_scores), xu32.numel(), 1 if add_one else 0, N.ptr(out))
    return out


def rank_of_pairs_by_user(scores, col_offset, begin, end, pair_indptr, xi32, target_scores, add_one=True):
    """rank_of_pairs for pairs grouped by user (pair_indptr int64 [n_users + 1] over the rows of ``scores``): a user's
    row is read once for all of the user's targets."""
    n_pairs = xi32.numel()
    out = torch.empty((n_pairs,), dtype=torch.int32, device=scores.device)
    N.call("trec_rank_of_pairs_by_user", N.ptr(scores), scores.stride(0), col_offset, begin, end, N.ptr(pair_indptr),
           N.ptr(xi32), N.ptr(target_scores), scores.shape[0], n_pairs, 1 if add_one else 0, N.ptr(out))
    return out


def sample_items(n_users, n_items, n_sampled, replace, seed, step, device="cuda", user_base=0):
    out = torch.empty((n_users, n_sampled), dtype=torch.int32, device=device)
    with _timed("sample_items"):
        N.call("trec_sample_items", n_users, int(user_base), n_items, n_sampled, 1 if replace else 0,
               int(seed) & (2 ** 64 - 1), int(step) & 0xFFFFFFFF, N.ptr(out))
    return out


def adam_tf_step(w, m, v, grad, lr_t, l2_coef, beta1=0.9, beta2=0.999, eps=1e-8):
    with _timed("adam_tf_step"):
        N.call("trec_adam_tf_step", N.ptr(w), N.ptr(m), N.ptr(v), N.ptr(_f32c(grad)), w.numel(), float(lr_t), beta1, beta2,
               eps, float(l2_coef))


# ------------------------------------------------------------------------------------------------ schedule on the device
def adam_schedule_advance(state, learning_rate, beta1=0.9, beta2=0.999, bump_sample_step=False):
    N.call("trec_adam_schedule_advance", N.ptr(state), float(learning_rate), float(beta1), float(beta2),
           1 if bump_sample_step else 0)


def adam_tf_step_dev(w, m, v, grad, state, l2_coef, beta1=0.9, beta2=0.999, eps=1e-8):
    N.call("trec_adam_tf_step_dev", N.ptr(w), N.ptr(m), N.ptr(v), N.ptr(_f32c(grad)), w.numel(), N.ptr(state),
           float(beta1), float(beta2), float(eps), float(l2_coef))


def sample_items_dev(n_users, n_items, n_sampled, replace, seed, state, device="cuda", user_base=0):
    """trec_sample_items with the step read from word 3 of the schedule state (uint32 bits)."""
    import ctypes
    out = torch.empty((n_users, n_sampled), dtype=torch.int32, device=device)
    step_ptr = ctypes.c_void_p(state.data_ptr() + 12)
    N.call("trec_sample_items_dev", n_users, int(user_base), n_items, n_sampled, 1 if replace else 0,
           int(seed) & (2 ** 64 - 1), step_ptr, N.ptr(out))
    return out
