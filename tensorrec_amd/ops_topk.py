"""
ops_topk -- K2: the user x item contraction and what consumes it without materialising [n_users, n_items]: operand preparation,
dense score slabs, the fused / two-stage exact top-k, the exact fp32 top-k through the bf16 filter (K2f), the int8 -> bf16 -> fp32
cascade with candidate lists and the pre-refined threshold (K2q / K2c), the wide-k and Euclidean routes, K5 merge.
The public module is tensorrec_amd.ops, which re-exports everything here.
"""
from .ops_base import (DTYPE_BF16, DTYPE_F32, MODE_DOT, MODE_EUCLIDEAN, N, _f32c, _timed, gemm_raw, group_pairs_by_item,
                       l2_normalize_rows, pair_scores_exact, rank_rows, torch)

# ------------------------------------------------------------------------------------------------ K2 / K4 / K7 / K8
def score_kpad(d):
    k = N.query("trec_score_kpad", int(d))
    if k < 0:
        raise ValueError("n_components = %d is beyond the MFMA score kernel's limit of 256" % d)
    return k


def score_prep(repr_, dtype=DTYPE_F32, normalize=False, want_sqnorm=False):
    """fp32 [n, d] representation -> MFMA operand [n, kpad] (fp32 or bf16), (squared row norms or None)."""
    x = _f32c(repr_.detach())
    n, d = x.shape
    kpad = score_kpad(d)
    if dtype == DTYPE_F32 and kpad == d and not normalize and not want_sqnorm:
        return x, None, kpad
    out = torch.empty((n, kpad), dtype=torch.float32 if dtype == DTYPE_F32 else torch.bfloat16, device=x.device)
    sq = torch.empty((n,), dtype=torch.float32, device=x.device) if want_sqnorm else None
    N.call("trec_score_prep", N.ptr(x), n, d, kpad, 1 if normalize else 0, dtype, N.ptr(out), N.ptr(sq))
    return out, sq, kpad


SCORE_KMAX = 256          # the MFMA score kernels keep one operand resident in registers: n_components <= 256


def dense_scores(user_repr, item_repr, dtype, normalize=False, mode=MODE_DOT, user_bias=None, item_bias=None, out=None):
    """[n_users, n_items] scores of a built-in prediction graph (+ biases) from the two representations: the MFMA score
    kernel with its fused epilogue for n_components <= 256; wider models take the K-looped fp32 MFMA GEMM
    (trec_gemm_f32) followed by elementwise passes -- slower per score, same fp32 semantics
    (prediction_graphs.py:49-50, :64-69, :84-100; recommendation_graphs.py:41)."""
    d = user_repr.shape[1]
    if d <= SCORE_KMAX:
        want_sq = mode == MODE_EUCLIDEAN
        u_op, u_sq, kpad = score_prep(user_repr, dtype, normalize=normalize, want_sqnorm=want_sq)
        i_op, i_sq, _ = score_prep(item_repr, dtype, normalize=normalize, want_sqnorm=want_sq)
        return score_store(u_op, i_op, dtype, kpad, user_bias, item_bias, mode, u_sq, i_sq, out=out)
    u, v = _f32c(user_repr.detach()), _f32c(item_repr.detach())
    if normalize:
        u, v = l2_normalize_rows(u).detach(), l2_normalize_rows(v).detach()
    s = gemm_raw(u, v, trans_b=True)
    if mode == MODE_EUCLIDEAN:
        r_u, r_v = (u * u).sum(dim=1, keepdim=True), (v * v).sum(dim=1, keepdim=True)
        s = -1.0 * torch.sqrt(torch.clamp((r_u - 2.0 * s) + r_v.t(), min=1e-16))
    if user_bias is not None:
        s = s + user_bias.reshape(-1, 1)
    if item_bias is not None:
        s = s + item_bias.reshape(1, -1)
    if out is not None:
        out.copy_(s)
        return out
    return s


TOPK_THRESHOLD_MIN_ITEMS = 8192       # from this row length on topk_from_scores selects by the k-th value instead of ranking the row


def topk_from_scores(scores, k):
    """(values [n_users, k], item ids int32 [n_users, k]) of a score slab in rank_predictions' order (value desc, index
    asc): exact ranks (rank_rows) select the entries -- the route of models the fused top-k kernels do not cover."""
    n_u, n_i = scores.shape
    kk = min(int(k), n_i)
    if n_i >= TOPK_THRESHOLD_MIN_ITEMS and 1 <= kk <= 256 and n_u >= 1:
        # long rows, few places: the k-th largest VALUE of a row does not depend on any tie rule (a selection, not a sort of the
        # row), the entries reaching it are k plus its ties, and trec_topk_merge orders them (value desc, index asc).  Rows with
        # NaN, a -inf k-th value or more than 1,024 such entries keep the exact-rank form below.
        # The selecting value need not BE the k-th largest: any t <= it keeps a superset whose k best are the row's.  First try the
        # k-th largest of the maxima of 512-entry blocks (k blocks hold an entry >= t each): ONE streaming pass over the slab, then
        # only the blocks whose maximum reaches t are looked at again (a few dozen per row) -- where a row-wise selection of the
        # k-th value, the comparison and the compaction of a [rows, 1M] mask cost ten times that.  Rows with heavy ties overflow
        # the limits there and take the exact k-th value.
        dev = scores.device
        bounds = (["blocks"] if n_i // 512 >= 2 * kk and N.load().trec_get_tuning(b"topk_slab_block_bound", 1) != 0 else []) + ["exact"]
        for bound in bounds:
            if bound == "blocks":
                n_full = n_i // 512 * 512
                bm = scores[:, :n_full].unflatten(1, (-1, 512)).amax(dim=2)
                if n_full < n_i:
                    bm = torch.cat([bm, scores[:, n_full:].amax(dim=1, keepdim=True)], dim=1)
                kth = torch.topk(bm, kk, dim=1, sorted=True).values[:, kk - 1:kk]
                bmask = bm >= kth
                n_blk = bmask.sum()
                if not bool((torch.isfinite(kth).all() & ~torch.isnan(bm).any() & (n_blk <= (4 * kk + 8) * n_u) &
                             (n_blk * 512 <= (1 << 28))).item()):
                    continue                                                    # (NaN / -inf rows, or ties across many blocks)
                rows_b, blk = torch.nonzero(bmask, as_tuple=True)              # row-major: rows ascending, blocks ascending
                cols_b = blk.reshape(-1, 1) * 512 + torch.arange(512, device=dev).reshape(1, -1)
                inside = cols_b < n_i                                           # (the ragged last block)
                vals_b = scores[rows_b.reshape(-1, 1), cols_b.clamp(max=n_i - 1)]
                pr, pc = torch.nonzero((vals_b >= kth[rows_b]) & inside, as_tuple=True)
                rows, cols, picked = rows_b[pr], cols_b[pr, pc], vals_b[pr, pc]
                cnt = torch.bincount(rows, minlength=n_u)
            else:
                kth = torch.topk(scores, kk, dim=1, sorted=True).values[:, kk - 1:kk]
                mask = scores >= kth
                cnt = mask.sum(dim=1)
                rows = None
            if not bool((torch.isfinite(kth).all() & (cnt.max() <= 1024) & (cnt.min() >= kk)).item()):
                continue
            if rows is None:
                rows, cols = torch.nonzero(mask, as_tuple=True)
                picked = scores[rows, cols]
            first = torch.cumsum(cnt, 0) - cnt
            slot = torch.arange(rows.numel(), device=dev) - first[rows]
            width = max(int(cnt.max().item()), kk)
            cv = torch.full((n_u, width), float('-inf'), dtype=torch.float32, device=scores.device)
            ci = torch.full((n_u, width), -1, dtype=torch.int32, device=scores.device)
            cv[rows, slot] = picked
            ci[rows, slot] = cols.to(torch.int32)
            mv, mi = topk_merge(cv, ci, kk)
            if kk == int(k):
                return mv, mi
            vals = torch.full((n_u, int(k)), float('-inf'), dtype=torch.float32, device=scores.device)
            idx = torch.full((n_u, int(k)), -1, dtype=torch.int32, device=scores.device)
            vals[:, :kk], idx[:, :kk] = mv, mi
            return vals, idx
    ranks = rank_rows(scores)
    rows, cols = torch.nonzero(ranks <= kk, as_tuple=True)
    pos = (ranks[rows, cols] - 1).long()
    vals = torch.full((n_u, int(k)), float('-inf'), dtype=torch.float32, device=scores.device)
    idx = torch.full((n_u, int(k)), -1, dtype=torch.int32, device=scores.device)
    vals[rows, pos] = scores[rows, cols]
    idx[rows, pos] = cols.to(torch.int32)
    return vals, idx


def score_store(users_op, items_op, dtype, kpad, user_bias=None, item_bias=None, mode=MODE_DOT, user_sq=None,
                item_sq=None, variant=0, out=None):
    n_u, n_i = users_op.shape[0], items_op.shape[0]
    if out is None:
        out = torch.empty((n_u, n_i), dtype=torch.float32, device=users_op.device)
    N.call("trec_score_gemm_store", N.ptr(users_op), N.ptr(items_op), dtype, kpad, n_u, n_i, N.ptr(user_bias),
           N.ptr(item_bias), mode, N.ptr(user_sq), N.ptr(item_sq), N.ptr(out), out.stride(0), variant)
    return out


def topk_chunks_for(n_users, dtype, kpad, n_items):
    """Item chunks so that the launch has >= ~2 workgroups per CU even for small user batches."""
    rows_wg = N.query("trec_score_rows_per_workgroup", dtype, kpad)
    rblocks = (n_users + rows_wg - 1) // rows_wg
    chunks = 1
    while rblocks * chunks < 1024 and chunks < 32 and n_items // (chunks * 2) >= 512:
        chunks *= 2
    return chunks


TWO_STAGE_MIN_ITEMS = 16384      # below this the direct fused kernel is cheaper than the extra passes
SUPERBLOCK_ROWS = 512


def score_topk(users_op, items_op, dtype, kpad, k, user_bias=None, item_bias=None, mode=MODE_DOT, user_sq=None,
               item_sq=None, item_index_base=0, n_chunks=None, variant=1, workspace=None, method="auto",
               floor_exchange=None):
    """Exact per-user top-k of the score matrix without materialising it.  Returns (values [U, k], item ids [U, k]),
    ordered (value desc, index asc).  ``method``: 'direct' (one fused pass with per-lane lists), 'two_stage'
    (superblock maxima -> select -> re-score, data-independent cost) or 'auto'."""
    if method == "auto":
        method = "two_stage" if items_op.shape[0] >= TWO_STAGE_MIN_ITEMS and n_chunks is None else "direct"
    if method == "two_stage":
        return score_topk_two_stage(users_op, items_op, dtype, kpad, k, user_bias, item_bias, mode, user_sq, item_sq,
                                    item_index_base, variant=variant, floor_exchange=floor_exchange)
    return score_topk_direct(users_op, items_op, dtype, kpad, k, user_bias, item_bias, mode, user_sq, item_sq,
                             item_index_base, n_chunks, variant, workspace)


def score_topk_direct(users_op, items_op, dtype, kpad, k, user_bias=None, item_bias=None, mode=MODE_DOT, user_sq=None,
                      item_sq=None, item_index_base=0, n_chunks=None, variant=1, workspace=None):
    cap = N.query("trec_score_topk_capacity", int(k))
    if cap < 0:
        raise ValueError("fused top-k supports k <= 16 (got %d)" % k)
    n_u, n_i = users_op.shape[0], items_op.shape[0]
    if n_chunks is None:
        n_chunks = topk_chunks_for(n_u, dtype, kpad, n_i)
    n_parts = N.query("trec_score_topk_parts", dtype, kpad, n_i, n_chunks)
    if workspace is None:
        pv = torch.empty((n_u, n_parts, cap), dtype=torch.float32, device=users_op.device)
        pi = torch.empty((n_u, n_parts, cap), dtype=torch.int32, device=users_op.device)
    else:
        pv, pi = workspace
    with _timed("score_gemm_topk"):
        N.call("trec_score_gemm_topk", N.ptr(users_op), N.ptr(items_op), dtype, kpad, n_u, n_i, item_index_base,
               N.ptr(user_bias), N.ptr(item_bias), mode, N.ptr(user_sq), N.ptr(item_sq), n_chunks, cap, N.ptr(pv),
               N.ptr(pi), variant)
    return topk_merge(pv.reshape(n_u, n_parts * cap), pi.reshape(n_u, n_parts * cap), k)


def score_topk_two_stage(users_op, items_op, dtype, kpad, k, user_bias=None, item_bias=None, mode=MODE_DOT,
                         user_sq=None, item_sq=None, item_index_base=0, sb_rows=None, variant=1, n_chunks=None,
                         floor_exchange=None):
    """See include/tensorrec_hip.h ("Two-stage exact top-k") and csrc/topk2.hip for the exactness argument.
    ``floor_exchange``: for item shards, a callable ``(sel_max [k, n_users]) -> floor [n_users]`` giving a lower bound
    of every user's GLOBAL k-th best score (sharding.shared_topk_floor: all-gather of the selected superblock maxima +
    their k-th largest); superblocks below it are not re-scored, so stage 3 costs ~k superblocks per user over all
    shards together instead of k per shard.  The lists returned are then exact only after the cross-shard merge."""
    cap = N.query("trec_score_topk_capacity", int(k))
    if cap < 0:
        raise ValueError("fused top-k supports k <= 16 (got %d)" % k)
    dev = users_op.device
    n_u, n_i = users_op.shape[0], items_op.shape[0]
    sb_rows = int(sb_rows or SUPERBLOCK_ROWS)
    n_sb = (n_i + sb_rows - 1) // sb_rows
    ksel = min(int(k), n_sb)
    rows_wg = N.query("trec_score_rows_per_workgroup", dtype, kpad)
    if n_chunks is None:
        # Stage 1 has no per-chunk state, so item chunks only set the workgroup count: aim for ~50 "rounds" of the
        # 512 co-resident workgroups (2 per CU) so that the last, partially filled round costs ~1% (measured at 1M x 1M:
        # 7 chunks = 27 rounds 171.1 ms, 14 chunks 169.0, 28 chunks 169.3, 56 chunks 169.2).
        rblocks = (n_u + rows_wg - 1) // rows_wg
        n_chunks = max(1, min(n_sb, -(-32 * 768 // rblocks)))
    # ---- stage 1: superblock maxima
    stride = (n_u + 3) // 4 * 4                     # 16-byte aligned rows: the tiled selection kernel's float4 loads
    blockmax = torch.empty((n_sb, stride), dtype=torch.float32, device=dev)
    with _timed("score_gemm_blockmax"):
        N.call("trec_score_gemm_blockmax", N.ptr(users_op), N.ptr(items_op), dtype, kpad, n_u, n_i, N.ptr(user_bias),
               N.ptr(item_bias), mode, N.ptr(user_sq), N.ptr(item_sq), sb_rows, n_chunks, N.ptr(blockmax), stride, variant)
    # ---- stage 2: the ksel best superblocks of every user
    sel = torch.empty((n_u, ksel), dtype=torch.int32, device=dev)
    tau = torch.empty((n_u,), dtype=torch.float32, device=dev) if ksel == int(k) else None   # floor needs k superblocks
    sel_max = None
    if floor_exchange is not None:                  # rows >= ksel stay -inf (a shard with fewer than k superblocks)
        sel_max = torch.full((int(k), n_u), float('-inf'), dtype=torch.float32, device=dev)
    with _timed("topk_select_blocks"):
        N.call("trec_topk_select_blocks", N.ptr(blockmax), n_sb, n_u, stride, ksel, N.ptr(sel), N.ptr(sel_max), N.ptr(tau))
    del blockmax
    floor = None
    if floor_exchange is not None:
        floor = tau = floor_exchange(sel_max).contiguous()
        sel_max = sel_max[:ksel]
    # ---- stage 3a: group (user, slot) pairs by superblock, pad groups to whole workgroups, gather operand rows
    n_pairs = n_u * ksel
    keys = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
    N.call("trec_topk_group_keys", N.ptr(sel), N.ptr(sel_max), N.ptr(floor), n_pairs, ksel, n_sb, N.ptr(keys))
    indptr_t, users_t, perm_t = group_pairs_by_item(None, keys, ksel, n_sb + 1)
    cnt_pad = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
    N.call("trec_topk_pad_counts", N.ptr(indptr_t), n_sb, rows_wg, N.ptr(cnt_pad))
    pstart = torch.empty((n_sb + 2,), dtype=torch.int64, device=dev)
    ws64 = torch.empty(((n_sb + 1 + 1023) // 1024 + 1,), dtype=torch.int64, device=dev)
    N.call("trec_exclusive_scan_i32", N.ptr(cnt_pad), n_sb + 1, N.ptr(ws64), N.ptr(pstart))
    max_rows = (n_pairs + min(n_sb, n_pairs) * (rows_wg - 1) + rows_wg - 1) // rows_wg * rows_wg
    g_op = torch.empty((max_rows, kpad), dtype=users_op.dtype, device=dev)
    g_bias = torch.empty((max_rows,), dtype=torch.float32, device=dev) if user_bias is not None else None
    g_sq = torch.empty((max_rows,), dtype=torch.float32, device=dev) if user_sq is not None else None
    g_tau = torch.empty((max_rows,), dtype=torch.float32, device=dev) if tau is not None else None
    row_pair = torch.empty((max_rows,), dtype=torch.int32, device=dev)
    rblock_chunk = torch.empty((max_rows // rows_wg,), dtype=torch.int32, device=dev)
    with _timed("topk_fill_groups"):
        N.call("trec_topk_fill_groups", N.ptr(pstart), N.ptr(indptr_t), N.ptr(users_t), N.ptr(perm_t), n_sb, rows_wg,
               max_rows, N.ptr(users_op), kpad * users_op.element_size(), N.ptr(user_bias), N.ptr(user_sq), N.ptr(tau),
               N.ptr(g_op), N.ptr(g_bias), N.ptr(g_sq), N.ptr(g_tau), N.ptr(row_pair), N.ptr(rblock_chunk))
    # ---- stage 3b: re-score the selected superblocks (every pair is written exactly once: ksel <= n_sb)
    pv = torch.empty((n_pairs * 2, cap), dtype=torch.float32, device=dev)
    pi = torch.full((n_pairs * 2, cap), -1, dtype=torch.int32, device=dev)      # unwritten lists read as empty
    with _timed("score_gemm_topk_grouped"):
        N.call("trec_score_gemm_topk_grouped", N.ptr(g_op), N.ptr(items_op), dtype, kpad, max_rows, n_i, item_index_base,
               N.ptr(g_bias), N.ptr(item_bias), mode, N.ptr(g_sq), N.ptr(item_sq), sb_rows, N.ptr(rblock_chunk),
               N.ptr(row_pair), N.ptr(g_tau), cap, N.ptr(pv), N.ptr(pi), variant & 1, None)
    # ---- stage 4: merge the ksel * 2 lists of every user
    return topk_merge(pv.reshape(n_u, ksel * 2 * cap), pi.reshape(n_u, ksel * 2 * cap), k)


# ------------------------------------------------------------------------------------------------ K2f: exact top-k, bf16 filter
# superblocks a user may keep (1M x 1M, d = 128, k = 10: 15.5 on average; 19 users of 1M need more than 32, none more than
# 48 -- a user beyond the limit goes to the exact fp32 fallback, whose launch chain costs ~1.2 ms however few users)
FILTER_KSEL = 48
FILTER_ONE_PASS_MIN_ENTRIES = 1 << 28   # table entries from which select + collect run as one scan (1 GB of maxima)
FILTER_CANDIDATES = 128    # candidates per user of the one-pass scan (entries above the provisional floor: ~45 at 1M x 1M)
FILTER_KSEL_WIDE = 320     # ... in the wide second pass over the flagged users (its finish kernel has no survivor limit)
COLLECT_KSEL_MAX = 4096    # slots per user trec_topk_collect_blocks can fill (csrc/topk2.hip)
WIDE_TIER2_MAX_FRACTION = 0.05   # the all-superblocks tier runs only when at most this fraction of the users is still flagged
LAST_FILTER_STATS = {}    # diagnostics of the most recent score_topk_filtered call (bench.py reports them)
FILTER_DEBUG = None       # diagnostics only: set to a dict to collect per-stage counters (each costs a host sync)
CANDIDATE_STATS = False   # diagnostics only: True adds "candidates_per_user" to LAST_FILTER_STATS (a reduction + a host read per call)


def _debug_counts(name, count):
    c = count.float()
    qs = torch.quantile(c[torch.randint(0, c.numel(), (min(c.numel(), 1_000_000),), device=c.device)],
                        torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=c.device))
    FILTER_DEBUG[name + "_mean"] = float(c.mean().item())
    FILTER_DEBUG[name + "_q50_90_99_999_max"] = [float(v) for v in qs]


class FilterOperand(object):
    """One side of the filtered top-k (trec_score_prep_filter): ``bf16`` [n, kpad] stage-1 / stage-3 operand, ``f32``
    [n, kpad] exact operand (the representation itself when it needs neither padding nor normalising), ``stats`` [n, 2]
    = {||x||, ||x - bf16(x)||} per row, ``gstats`` [3] = maxima of both and of |bias| over the rows (item side).
    The int8 pre-filter (score_prep_i8_pair) adds ``i8`` [n, kpad] int8, ``stats8`` [n, 2] = {||x||, ||x - scale q||},
    and on the item side ``bias_q`` int32 [n], ``sb_stats`` [n_sb, 4] = per superblock of ``sb_rows`` items {scale, max
    ||y|| + ||dy||, max ||dy||, max bias quantisation error}, ``gstats8`` [4] ([2] = max |bias|), ``scales`` [3] ([0] = user scale).

    Users sorted by int8 scale class (score_prep_filter(sort_users=True) -> trec_user_prep_sorted, csrc/user_prep.hip): every
    array is in LAYOUT order, ``n`` = the layout's rows (a bound the host knows: n_real + up to 32 x (wg_rows - 1) padding
    rows), ``src`` int32 [n] = the caller's row of a layout row (-1: none -- such rows are zero and keep nothing), ``pos`` int32
    [n_real] = the layout row of a caller's row, ``wg_scale`` / ``wg_class`` per int8 workgroup of ``wg_rows`` rows (scale 0 = an
    idle workgroup beyond the padded rows), ``ladder`` [64] the classes' scales, ``class_used`` [64], ``gmax`` [1], ``meta`` int32
    [2] = {padded rows, n_real} (on the device: nothing here is read by the host), ``bias_sorted`` the user bias in layout order
    when it was given to the preparation (``bias_ref``)."""
    __slots__ = ("bf16", "f32", "n", "d", "kpad", "stats", "gstats", "i8", "stats8", "bias_q", "gstats8", "scales",
                 "sb_stats", "sb_rows", "cascade_too_loose", "gmax", "wg_scale", "wg_class", "wg_rows", "pos",
                 "src", "n_real", "ladder", "class_used", "meta", "bias_sorted", "bias_ref", "bias_owner", "__weakref__")

    def __init__(self):
        self.i8 = self.stats8 = self.bias_q = self.gstats8 = self.scales = self.sb_stats = self.sb_rows = None
        self.cascade_too_loose = False      # set on the item side when the int8 bound did not pay for this catalogue
        self.gmax = self.wg_scale = self.wg_class = self.wg_rows = self.pos = self.src = self.n_real = None
        self.ladder = self.class_used = self.meta = self.bias_sorted = self.bias_ref = None
        self.gstats = None
        self.bias_owner = None              # item side: weak reference to the USER operand its integer biases were derived for

    # ---- diagnostics / tests only (each is a torch op or a host sync; nothing on the product path reads them)
    @property
    def perm(self):
        """int64 [n]: the caller's row behind every layout row (rows without one read row 0), or None for unsorted operands."""
        return None if self.src is None else self.src.clamp(min=0).long()

    @property
    def pad(self):
        """bool [n]: layout rows without a source, or None."""
        return None if self.src is None else self.src < 0

    @property
    def order(self):
        """int64 [n_real]: the caller's rows in layout order."""
        return None if self.src is None else self.src[self.src >= 0].long()


I8_CLASSES_PER_OCTAVE = 4         # user scale classes: a geometric ladder below the largest wanted scale, 2^(1/4) apart
I8_N_CLASSES = 64                 # ... over 16 octaves; smaller rows share the last class


def i8_user_classes_enabled():
    return N.load().trec_get_tuning(b"i8_user_classes", 1) != 0


I8_CLASS_BAND = 2                 # classes per band: the users of one int8 workgroup come from ONE band (scales within 2^(1/2))


def zero_block(n_words, device):
    """int32 [n_words] of zeros from ONE hipMemsetAsync (trec_fill_zero): the counters and running maxima a call starts from
    are slices of such a block instead of a torch.zeros launch each."""
    buf = torch.empty((int(n_words),), dtype=torch.int32, device=device)
    N.call("trec_fill_zero", N.ptr(buf), int(n_words) * 4)
    return buf


def _prep_users_sorted(x, normalize, k, user_bias=None):
    """The user side of the cascade in five launches and no host read (csrc/user_prep.hip): scale class per row, stable
    counting sort by class, bands of I8_CLASS_BAND classes padded to whole int8 workgroups, one gather pass that writes the
    fp32 / bf16 / int8 operands with both error norms (and the user bias) in layout order."""
    n, d = x.shape
    kpad = score_kpad(d)
    dev = x.device
    wg_rows = int(N.query("trec_score_blockmax_i8_rows_per_workgroup", 10 if int(k) <= 10 else 16))
    n_alloc = int(N.query("trec_user_prep_alloc_rows", n, wg_rows))
    n_wg = n_alloc // wg_rows
    ws_bytes = int(N.query("trec_user_prep_workspace_bytes", n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    small = torch.empty((8 + 2 * I8_N_CLASSES + 2 * n_wg,), dtype=torch.int32, device=dev)
    N.call("trec_fill_zero", N.ptr(small), 32)                           # gmax (a running maximum) | meta
    op = FilterOperand()
    op.gmax = small[0:1].view(torch.float32)
    op.meta = small[4:6]
    op.ladder = small[8:8 + I8_N_CLASSES].view(torch.float32)
    op.class_used = small[8 + I8_N_CLASSES:8 + 2 * I8_N_CLASSES]
    op.wg_scale = small[8 + 2 * I8_N_CLASSES:8 + 2 * I8_N_CLASSES + n_wg].view(torch.float32)
    op.wg_class = small[8 + 2 * I8_N_CLASSES + n_wg:]
    op.n, op.n_real, op.d, op.kpad, op.wg_rows = n_alloc, n, d, kpad, wg_rows
    op.src = torch.empty((n_alloc,), dtype=torch.int32, device=dev)
    op.pos = torch.empty((n,), dtype=torch.int32, device=dev)
    op.f32 = torch.empty((n_alloc, kpad), dtype=torch.float32, device=dev)
    op.bf16 = torch.empty((n_alloc, kpad), dtype=torch.bfloat16, device=dev)
    op.i8 = torch.empty((n_alloc, kpad), dtype=torch.int8, device=dev)
    op.stats = torch.empty((n_alloc, 2), dtype=torch.float32, device=dev)
    op.stats8 = torch.empty((n_alloc, 2), dtype=torch.float32, device=dev)
    ub = _f32c(user_bias.detach()).reshape(-1) if user_bias is not None else None
    op.bias_sorted = torch.empty((n_alloc,), dtype=torch.float32, device=dev) if ub is not None else None
    op.bias_ref = user_bias
    with _timed("user_prep_sorted"):
        N.call("trec_user_prep_sorted", N.ptr(x), n, d, kpad, 1 if normalize else 0, N.ptr(ub), wg_rows, n_alloc, N.ptr(ws),
               ws_bytes, N.ptr(op.src), N.ptr(op.pos), N.ptr(op.wg_scale), N.ptr(op.wg_class), N.ptr(op.ladder),
               N.ptr(op.class_used), N.ptr(op.gmax), N.ptr(op.meta), N.ptr(op.f32), N.ptr(op.bf16), N.ptr(op.stats),
               N.ptr(op.i8), N.ptr(op.stats8), N.ptr(op.bias_sorted))
    return op


def score_prep_filter(repr_, normalize=False, bias=None, want_gstats=False, sort_users=False, k=10, user_bias=None):
    """Operands of the filtered top-k for one side (FilterOperand).  ``sort_users`` (the USER side of score_topk_filtered with
    the int8 pre-filter): the rows are laid out sorted by the int8 scale class each wants, bands of I8_CLASS_BAND adjacent
    classes padded to whole int8 workgroups, and the int8 operand is made in the same pass (_prep_users_sorted: no host
    read; the layout has ``op.n`` >= n rows, ``op.src`` / ``op.pos`` map between it and the caller's rows).  ``k``: the top-k
    this operand is for (the int8 kernel's workgroup covers 768 users for k <= 10, 512 above).  ``user_bias`` (with
    sort_users): the users' biases, also wanted in layout order -- pass the same tensor to score_topk_filtered."""
    x = _f32c(repr_.detach())
    n, d = x.shape
    kpad = score_kpad(d)
    if sort_users and kpad <= 128 and n > 0 and i8_user_classes_enabled():
        return _prep_users_sorted(x, normalize, k, user_bias)
    op = FilterOperand()
    op.n, op.d, op.kpad = n, d, kpad
    own_f32 = normalize or kpad != d
    op.f32 = torch.empty((n, kpad), dtype=torch.float32, device=x.device) if own_f32 else x
    op.bf16 = torch.empty((n, kpad), dtype=torch.bfloat16, device=x.device)
    op.stats = torch.empty((n, 2), dtype=torch.float32, device=x.device)
    op.gstats = zero_block(4, x.device)[:3].view(torch.float32) if want_gstats else None
    with _timed("score_prep_filter"):
        N.call("trec_score_prep_filter", N.ptr(x), n, d, kpad, 1 if normalize else 0, N.ptr(bias),
               N.ptr(op.f32) if own_f32 else None, N.ptr(op.bf16), N.ptr(op.stats), N.ptr(op.gstats))
    return op


def spmm_filter_operand(features, w, bias=None, want_gstats=False):
    """K1 with the filtered top-k's operand as its epilogue (trec_spmm_csr_filter): representation = features . w in fp32
    (the exact operand) together with its bf16 image and rounding-error norms -- no separate pass over the representation.
    For representations that enter the score kernels unchanged: n_components in (32, 64, 128, 256), dot products."""
    w = _f32c(w)
    d = w.shape[1]
    n = features.shape[0]
    op = FilterOperand()
    op.n, op.d, op.kpad = n, d, d
    op.f32 = torch.empty((n, d), dtype=torch.float32, device=w.device)
    op.bf16 = torch.empty((n, d), dtype=torch.bfloat16, device=w.device)
    op.stats = torch.empty((n, 2), dtype=torch.float32, device=w.device)
    op.gstats = torch.zeros((3,), dtype=torch.float32, device=w.device) if want_gstats else None
    with _timed("spmm_csr"):
        N.call("trec_spmm_csr_filter", N.ptr(features.indptr), N.ptr(features.indices), N.ptr(features.values), n,
               features.nnz, N.ptr(w), d, N.ptr(op.f32), N.ptr(op.bf16), N.ptr(op.stats), N.ptr(op.gstats))
    if want_gstats and bias is not None:
        import ctypes
        N.call("trec_absmax", N.ptr(bias), bias.numel(), ctypes.c_void_p(op.gstats.data_ptr() + 8))
    return op


I8_USER_CLIP_SIGMAS = 4.0        # user rows clip at 4 rms (a clipped user only widens ITS bound; measured at 1M x 1M: refined
                                 # pairs 139M at 5.0, 125M at 4.5, 114M at 4.0, 119M at 3.5); item rows never clip
# refine at most this fraction of the (superblock, user) pairs; beyond it bf16 does it all.  Break-even: the int8 pass costs
# ~0.53 of a dense bf16 pass and a refined pair ~1.5x a dense one (gathered rows), so the cascade wins below ~0.31 and still
# beats "int8 pass wasted + dense bf16" up to ~0.67 (measured, 32,768 x 1M: 23.6% refined 9.3 ms vs 8.1 bf16-only vs 15.0 wasted)
CASCADE_MAX_REFINED = 0.45
CASCADE_ROW_CAPACITY = 0.50      # fixed capacity of a superblock's user list (fraction of the users); fuller rows are "hot":
                                 # the dense kernel re-scores them for everybody at 1.5x the grouped kernel's rate
CASCADE_MAX_HOT = 1 << 20        # superblocks that may be hot (no limit of its own: CASCADE_MAX_REFINED bounds the work)
CASCADE_CANDIDATES = 256         # candidate items per user the refining launches may list (trec_score_gemm_refine_candidates:
                                 # every item of a refined pair within eps of the k-th largest int8 lower bound; ~30 at 1M x 1M,
                                 # 138 (median) on clustered rows; the finish reads the first 64 unasked, the rest by the count)
CASCADE_DENSE_USER_LIMIT = 16    # of 32 sampled superblocks kept (Gaussian rows keep 2.6 % of them, clustered ones 23 %): the int8 bound says nothing about this user -- flagged at once
CASCADE_MAX_CHUNKS = 128        # item chunks of the int8 / bf16 stage-1 launches at most: small user batches pay the selection's serial walk over
                                # chunks x k list rows (256 users x 1M items, k = 10: 1.41 ms per call at 256 chunks, 1.11 at 128, 1.24 at 64 where the
                                # int8 launch runs short of workgroups; tuning cascade_max_chunks, profiles/r06_small_batch_bench.json)
CASCADE_PREREFINE_MIN_SB = 32   # superblocks from which the users' k best superblocks are pre-refined (k / n_sb of all pairs; the product
                                # path runs the cascade from 512 superblocks on, the tests from 40)
CASCADE_MIN_ITEMS = 262144       # below ~512 superblocks the k-th largest maximum is not selective enough for int8 to pay


TOPK_USER_BATCH_MAX = 4_194_304     # users per pass at most, whatever the free memory says (int32 offsets inside the kernels' tables)


def topk_user_batch(n_users, n_items, n_components, device, fraction=0.6, route="cascade", k=10):
    """Users per pass of predict_top_k when the caller names no batch size: what ``fraction`` of the FREE device memory holds.
    ``route`` "cascade" (the exact top-k through the int8 / bf16 filters): per user a column of the superblock-maxima table (4 B
    per superblock), two half columns of user-list slots (CASCADE_ROW_CAPACITY x 4 B each: pre-refinement and compaction), the
    operands three times over (fp32 + bf16 + int8), ~1 KB of chunk lists and CASCADE_CANDIDATES x 8 B of candidate slots;
    "wide" (17 <= k <= 64): 1,024 candidate slots with their exact scores and masks.  Any other route (score_topk_two_stage: bf16
    precision, Euclidean with k > 12 or several tastes, the filters switched off): the table column, the operand once, the
    gathered operand of the k selected superblocks (k x kpad x 4 B) and the stage-2 lists (2 k parts x capacity x 8 B, twice) --
    12-20 KB per user at k = 16.  30 % on top for the allocator.  Never below 65,536 (the old fixed default) and
    never above TOPK_USER_BATCH_MAX."""
    n_sb = (int(n_items) + SUPERBLOCK_ROWS - 1) // SUPERBLOCK_ROWS
    kpad = max(32, (int(n_components) + 31) // 32 * 32)
    if route == "cascade":           # (two user lists per superblock since round 5: the pre-refinement's and the compaction's; round 6:
        # the pre-refinement's maxima in its list's layout -- a third half column -- and its list positions, 4 B per slot)
        per_user = 1.3 * (n_sb * (4 + 12 * CASCADE_ROW_CAPACITY) + 8 * kpad + 1024 + 8 * CASCADE_CANDIDATES + 256 + 16 * int(k))
    elif route == "wide":            # score_topk_filtered_wide: 1,024 candidate slots (8 B each) and the k places; the cascade's three
        # half columns of user-list slots (the wide route pre-refines too) and 16 B per pre-refined slot (superblock, position, value, place)
        per_user = 1.3 * (n_sb * (4 + 12 * CASCADE_ROW_CAPACITY) + 8 * kpad + 1024 + 8 * WIDE_CANDIDATES + 32 * int(k))
    else:
        # (the fused lists hold up to 16 entries: trec_score_topk_capacity is -1 beyond that, and k > 16 finishes through score
        # slabs of [users, SUPERBLOCK_ROWS * k] fp32 instead -- sized by the larger of the two)
        cap = max(16, int(N.query("trec_score_topk_capacity", min(16, int(k)))))
        lists = 2 * (2 * int(k)) * cap * 8 if int(k) <= 16 else max(2 * (2 * int(k)) * cap * 8, 2 * 4 * SUPERBLOCK_ROWS * int(k))
        per_user = 1.3 * (n_sb * 4 + 6 * kpad + int(k) * kpad * 4 + lists + 1024)
    try:
        free, _total = torch.cuda.mem_get_info(device)
    except Exception:                                   # pragma: no cover
        free = 16 << 30
    return int(max(65536, min(int(n_users), TOPK_USER_BATCH_MAX, fraction * free / per_user)))


def cascade_prefilter_for(n_components, n_items_total):
    """"int8" when the int8 pre-filter is worth trying for this shape (tuning ``topk_int8_prefilter``, default on), else None."""
    if N.load().trec_get_tuning(b"topk_int8_prefilter", 1) == 0:
        return None
    return "int8" if score_kpad(n_components) in (64, 128) and n_items_total >= CASCADE_MIN_ITEMS else None


def score_prep_i8_pair(uop, iop, item_bias=None, sb_rows=None, top_k=10):
    """int8 operands of both sides for the cascade's pre-filter, from the fp32 operands of FilterOperand (already
    normalised / padded).  Items: one scale per superblock of ``sb_rows`` rows (max |y| / 127 over the superblock: no item
    clips), with the superblock's error maxima in ``iop.sb_stats`` [n_sb, 4]; quantised once per ``iop``.
    Users sorted by scale class (``uop.src``, score_prep_filter(sort_users=True)): their int8 rows exist already -- every int8
    workgroup of ``wg_rows`` users has the scale of its first (largest) row on the ladder gmax * 2^(-c / 4) -- and the integer
    item biases are made once per class in use (``iop.bias_q`` [n_classes, n_items]).  Users in the caller's order: ONE
    scale, min(4 rms, max |x|) / 127 (``iop.scales[0]``).  A new batch of users only re-derives the item biases."""
    if uop.kpad > 128:
        raise ValueError("int8 pre-filter covers kpad <= 128")
    sb_rows = int(sb_rows or SUPERBLOCK_ROWS)
    dev = uop.f32.device
    classes = uop.src is not None
    with _timed("score_prep_i8"):
        fresh_items = iop.i8 is None or iop.sb_rows != sb_rows
        n_sb = (iop.n + sb_rows - 1) // sb_rows
        if fresh_items:
            zb = zero_block(8 + 4 * n_sb, dev).view(torch.float32)       # scales [3] | gstats8 [4] | sb_stats [n_sb, 4], one memset
            iop.scales = zb[0:3]
            iop.gstats8 = zb[4:8]
            iop.sb_stats = zb[8:].reshape(n_sb, 4)
            iop.sb_rows = sb_rows
            iop.i8 = torch.empty((iop.n, iop.kpad), dtype=torch.int8, device=dev)
            iop.stats8 = torch.empty((iop.n, 2), dtype=torch.float32, device=dev)
            N.call("trec_score_prep_i8", N.ptr(iop.f32), iop.n, iop.f32.shape[1], iop.kpad, 1, 0.0, sb_rows, None,
                   N.ptr(iop.scales), None, N.ptr(iop.i8), N.ptr(iop.stats8), None, N.ptr(iop.sb_stats), N.ptr(iop.gstats8))
        else:
            iop.gstats8[2:].zero_()
            iop.sb_stats[:, 3].zero_()
        if classes:
            wg_rows = int(N.query("trec_score_blockmax_i8_rows_per_workgroup", int(top_k)))
            if wg_rows != uop.wg_rows:
                raise ValueError("the user operand was laid out for int8 workgroups of %s rows, this call needs %d "
                                 "(score_prep_filter(sort_users=True, k=...) must be given the same k)" % (uop.wg_rows, wg_rows))
            iop.bias_q = None
            if item_bias is not None:
                iop.bias_q = torch.empty((I8_N_CLASSES, iop.n), dtype=torch.int32, device=dev)   # only the classes in use are touched
                N.call("trec_score_bias_i8_classes", N.ptr(item_bias), iop.n, sb_rows, N.ptr(uop.ladder), N.ptr(uop.class_used),
                       I8_N_CLASSES, N.ptr(iop.sb_stats), N.ptr(iop.bias_q), N.ptr(iop.gstats8))
        else:
            uop.i8 = torch.empty((uop.n, uop.kpad), dtype=torch.int8, device=dev)
            uop.stats8 = torch.empty((uop.n, 2), dtype=torch.float32, device=dev)
            uop.wg_class = uop.wg_scale = None
            ws = torch.empty((2,), dtype=torch.float64, device=dev)
            clip = N.load().trec_get_tuning(b"i8_user_clip_x10", int(I8_USER_CLIP_SIGMAS * 10)) / 10.0
            N.call("trec_score_prep_i8", N.ptr(uop.f32), uop.n, uop.f32.shape[1], uop.kpad, 0, float(clip), 0, None,
                   N.ptr(iop.scales), N.ptr(ws), N.ptr(uop.i8), N.ptr(uop.stats8), None, None, None)
            iop.bias_q = torch.empty((iop.n,), dtype=torch.int32, device=dev) if item_bias is not None else None
            if item_bias is not None:
                N.call("trec_score_prep_i8", None, iop.n, iop.f32.shape[1], iop.kpad, 2, 0.0, sb_rows, N.ptr(item_bias),
                       N.ptr(iop.scales), None, None, None, N.ptr(iop.bias_q), N.ptr(iop.sb_stats), N.ptr(iop.gstats8))
    import weakref
    iop.bias_owner = weakref.ref(uop)       # the integer item biases (and sb_stats[:, 3]) belong to THIS batch's user scales
    return uop, iop


def blockmax_i8_chunks(n_items, n_chunks, sb_rows):
    """(chunk length in items, number of chunks) trec_score_gemm_blockmax_i8 uses for a requested chunk count."""
    chunk_len = -(-(-(-n_items // n_chunks)) // sb_rows) * sb_rows
    return chunk_len, -(-n_items // chunk_len)


import contextlib as _contextlib


@_contextlib.contextmanager
def _tail_of(tail_stream, *tensors):
    """Launches inside run on ``tail_stream`` (None: the current one) once everything queued on the current stream so far is
    done; ``tensors`` -- allocated on the current stream, used inside -- are recorded on it so that the allocator does not hand
    their memory out again before the tail stream is through with them."""
    if tail_stream is None:
        yield
        return
    tail_stream.wait_stream(torch.cuda.current_stream())
    for t in tensors:
        if t is not None:
            t.record_stream(tail_stream)
    with torch.cuda.stream(tail_stream):
        yield


class _Pending(object):
    """A filtered top-k whose tail is still running on another stream: ``complete()`` -- once the caller's stream has waited
    for that stream -- reads the flagged-user counter, re-does those users and returns (values, ids, stats)."""
    __slots__ = ("complete",)

    def __init__(self, complete):
        self.complete = complete


class _Candidates(object):
    """What the refining launches listed (trec_score_gemm_refine_candidates): per user ``n`` appended entries of ``items``
    [n_users, cap, 2] = {item id, score bits}, made with the provisional floor ``floor0`` (+inf: nothing listed); ``flag`` /
    ``n_flagged``: users without a usable bound so far."""
    __slots__ = ("n", "items", "cap", "floor0", "flag", "n_flagged", "pre")


RESIDENT_SEGMENT = 2048          # users of a superblock's list one workgroup of the item-resident refining launch streams


def _refine_resident(sb_rows, kpad):
    """tuning refine_resident = 1: the refining launches keep the ITEMS resident (csrc/refine_resident.hip; the superblock must be
    the 512 items its four waves hold).  Measured on par with the user-resident kernel, not faster (1.92 + 2.70 against 1.81 + 2.73 ms
    at 1M x 1M, profiles/r06_refine_resident_ab.txt): the default stays the user-resident kernel, this one is the A/B alternative."""
    return int(sb_rows) == 512 and kpad in (64, 128) and N.load().trec_get_tuning(b"refine_resident", 0) != 0


def _resident_segments(rcap):
    seg = int(N.load().trec_get_tuning(b"refine_resident_seg", RESIDENT_SEGMENT))
    seg = max(64, seg // 64 * 64)
    return seg, (int(rcap) + seg - 1) // seg


def cascade_lists_candidates():
    """Tuning ``cascade_candidates`` (default 1): the bf16 refining launches of the cascade also list, per user, the items
    that can still reach the top-k, and trec_topk_candidates_finish ends the call -- no table scan, no grouping by superblock,
    no grouped list kernel.  0: the bf16 filter's tail runs on the mixed table (the round-2 form)."""
    return N.load().trec_get_tuning(b"cascade_candidates", 1) != 0


def _cascade_stage1(uop, iop, k, user_bias, item_bias, sb_rows, n_sb, n_chunks, floor_exchange, stats_exchange, gstats_all=None,
                    item_index_base=0, tail_stream=None, candidates_cap=None):
    """Stages 0-1 of the int8 -> bf16 -> fp32 cascade (csrc/topk_cascade.hip): the [n_sb, n_users] table of superblock
    maxima whose entries are bf16 maxima wherever a top-k item can be and int8 maxima elsewhere (None after an overflow), its
    row stride, (resident rows of the bf16 launches, overflow), the k-th largest int8 lower bounds, and -- the default, DESIGN
    5e -- the candidate lists the refining launches made (else None).  Every superblock has a list of CASCADE_ROW_CAPACITY of
    the users (only the workgroup slots that hold rows are launched); rows kept by more are "hot" and re-scored for everybody
    by a dense launch.  When the int8 bound is too loose for the data -- more than CASCADE_MAX_REFINED of all pairs wanted --
    nothing is refined and the caller runs the dense bf16 stage 1 instead: the ONE host read of the call (three int64: rows,
    overflow, hot rows) happens right after the compaction, before any bf16 launch.  ``tail_stream``: the refining launches
    go there (ops._score_topk_filtered_pipelined)."""
    dev = uop.bf16.device
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad
    kk = int(k)
    top_k = 10 if kk <= 10 else 16
    # (the item rows are quantised once per catalogue; the integer item biases are in units of the USERS' scale products, so
    # every user batch derives them again -- a class-sorted user operand brings its own int8 rows, which says nothing about
    # whose scales the item side's tables were made for)
    if uop.i8 is None or iop.i8 is None or iop.sb_rows != sb_rows or iop.bias_owner is None or iop.bias_owner() is not uop:
        score_prep_i8_pair(uop, iop, item_bias, sb_rows, top_k)
    gstats8 = iop.gstats8
    if stats_exchange is not None:                  # max |item bias| over ALL shards enters every user's bound: the same number
        gstats8 = gstats8.clone()                   # the bf16 filter's statistics carry (gstats[2]), already MAX-reduced
        gstats8[2] = gstats_all[2] if gstats_all is not None else stats_exchange(gstats8)[2]
    user_err = torch.empty((n_u, 4), dtype=torch.float32, device=dev)
    N.call("trec_score_user_err_i8", N.ptr(uop.stats8), N.ptr(user_bias), N.ptr(gstats8), kpad, n_u, N.ptr(iop.scales),
           N.ptr(uop.wg_scale), int(uop.wg_rows or 0) if uop.wg_scale is not None else 0, N.ptr(user_err))
    stride = (n_u + 3) // 4 * 4
    chunk_len, n_ch = blockmax_i8_chunks(n_i, n_chunks, sb_rows)
    table = torch.empty((n_sb, stride), dtype=torch.float32, device=dev)
    chunk_top = torch.empty((n_ch * top_k, stride), dtype=torch.float32, device=dev)
    lib = N.load()
    lists = cascade_lists_candidates() and gstats_all is not None and sb_rows <= 65536
    one_pass = lib.trec_get_tuning(b"cascade_rows_onepass", 1) != 0 and lib.trec_get_tuning(b"blockmax_bf16_mfma16", 1) != 0
    # PRE-REFINEMENT (csrc/topk_filter.hip): the k superblocks with a user's k largest lower bounds are refined first and
    # tau rises to min(their bf16 maxima) - eps -- one e closer to the k-th best score, so the compaction keeps about half as many
    # pairs and the lists half as many candidates.  The chunk lists then carry the superblock's index inside its chunk in their low
    # bits.  Single process, candidate lists, catalogues the LDS counters cover; item shards keep the exchanged tau8.
    sb_per_chunk = chunk_len // sb_rows
    pre = (one_pass and lists and floor_exchange is None and stats_exchange is None and kk <= WIDE_K_MAX
           and lib.trec_get_tuning(b"cascade_prerefine", 1) != 0 and sb_per_chunk <= 4096
           and CASCADE_PREREFINE_MIN_SB <= n_sb <= int(N.query("trec_topk_prerefine_max_superblocks")))
    if pre and kk > 16:
        # the wide route (17 <= k <= 64): only in the default form -- the listing + marking launch and the threshold read back from
        # its lists -- whose kernels take up to 64 slots per user (tuning wide_prerefine = 0: as before round 6, without)
        pre = (lib.trec_get_tuning(b"cascade_prerefine", 1) == 1 and lib.trec_get_tuning(b"prerefine_marked", 1) != 0
               and lib.trec_get_tuning(b"wide_prerefine", 1) != 0 and not _refine_resident(sb_rows, kpad)
               and n_sb >= 4 * kk)                          # (k of n_sb superblocks per user: a small share of the pairs only then)
    with _timed("score_gemm_blockmax_i8"):
        N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), kpad, n_u, n_i, N.ptr(user_bias),
               N.ptr(iop.bias_q), N.ptr(iop.scales), N.ptr(iop.sb_stats), sb_rows, n_chunks, N.ptr(table), stride,
               N.ptr(user_err), N.ptr(chunk_top), top_k | (0x100 if pre else 0), N.ptr(uop.wg_scale), N.ptr(uop.wg_class),
               int(uop.wg_rows or 0) if uop.wg_scale is not None else 0)
    # tau = the k-th largest LOWER bound: from the chunks' lists (k rows per chunk), not from the 7.8 GB table
    sel = torch.empty((n_u, kk), dtype=torch.int32, device=dev)
    sel_max = torch.empty((kk, n_u), dtype=torch.float32, device=dev) if (floor_exchange is not None or pre) else None
    tau = torch.empty((n_u,), dtype=torch.float32, device=dev)
    with _timed("topk_select_blocks"):
        if kk <= 16:
            N.call("trec_topk_select_blocks", N.ptr(chunk_top), n_ch * top_k, n_u, stride, kk, N.ptr(sel), N.ptr(sel_max),
                   N.ptr(tau))
        else:
            # k up to 64 (score_topk_filtered_wide): the k-th largest entry of the UNION of the chunks' 16-entry lists -- at most the
            # k-th largest lower bound of the column, and still k distinct superblocks that each hold an item at or above it
            N.call("trec_topk_select_blocks_ex", N.ptr(chunk_top), n_ch * top_k, n_u, stride, kk, kk, N.ptr(sel), N.ptr(sel_max),
                   N.ptr(tau))
    if floor_exchange is not None:
        tau = floor_exchange(sel_max).contiguous()
    # the counters this call starts from, as ONE zeroed block: status int64 [3] | n_flagged | row_count [n_sb]
    zb = zero_block(8 + n_sb, dev)
    status = zb[0:6].view(torch.int64)
    n_flagged0 = zb[6:7]
    row_count = zb[8:8 + n_sb]
    cands = None
    if one_pass and lists:
        # the refining launches also list every item that can still reach the top-k: provisional floor = the k-th
        # largest int8 lower bound less ONE eps of the bf16 filter.  ONE pass over the users (trec_topk_cascade_floor) makes the
        # thresholds: rows without a source keep nothing (tau = floor = +inf), users without a usable bound are flagged and
        # list nothing, the list counters start at zero
        cands = _Candidates()
        cands.pre = None
        # (an argument, not a process-global knob flipped around the call: another thread's predict_top_k must not see the wide
        # route's 1,024 slots; the tuning knob stays for A/B runs of the default)
        cands.cap = int(candidates_cap) if candidates_cap is not None else \
            int(N.load().trec_get_tuning(b"cascade_candidates_cap", CASCADE_CANDIDATES))
        cands.floor0 = torch.empty((n_u,), dtype=torch.float32, device=dev)
        cands.flag = torch.empty((n_u,), dtype=torch.int32, device=dev)
        cands.n_flagged = n_flagged0
        cands.n = torch.empty((n_u,), dtype=torch.int32, device=dev)
        N.call("trec_topk_cascade_floor", N.ptr(tau), N.ptr(uop.src), N.ptr(uop.stats), N.ptr(user_bias), N.ptr(gstats_all), kpad,
               n_u, N.ptr(cands.floor0), N.ptr(cands.flag), N.ptr(cands.n_flagged), N.ptr(cands.n))
        try:
            cands.items = torch.empty((n_u, cands.cap, 2), dtype=torch.int32, device=dev)     # only the listed part is touched
        except torch.cuda.OutOfMemoryError:
            # (2 KB of list slots per user is a reservation, not traffic; on a device that cannot spare it the lists shrink
            # to 64 slots -- users beyond them are flagged and re-done on their table column -- ADVICE r3)
            cands.cap = 64
            cands.items = torch.empty((n_u, cands.cap, 2), dtype=torch.int32, device=dev)
    if pre:
        # ---- the pre-refinement: lists of the users' k best superblocks, the bf16 launch over them, the sharper threshold.
        # Default (tuning cascade_prerefine = 1): that launch also LISTS the candidates of those pairs (with the provisional floor
        # tau8 - eps) and the pairs then leave the compaction's sight (table entry -inf, the bf16 maxima saved) -- nothing is refined
        # twice.  cascade_prerefine = 2: maxima only, the entries marked +inf and refined again by the listing launch (A/B).
        listed = 1 if lib.trec_get_tuning(b"cascade_prerefine", 1) == 1 else 0
        n_cap_a = int(uop.n_real or n_u)
        if listed:
            # the compaction's own capacity (half of the users per superblock: on fitted / Zipf catalogues the k best superblocks of
            # most users are the same few popular ones) -- only the workgroup slots that hold rows are launched (wg_map below)
            rcap_frac_a = lib.trec_get_tuning(b"cascade_rcap_pct", int(100 * CASCADE_ROW_CAPACITY)) / 100.0
            rcap_a = (int(rcap_frac_a * n_cap_a) + 511) // 512 * 512 + 512
        else:
            rcap_a = (2 * (n_cap_a * kk // n_sb + 1) + 1024 + 511) // 512 * 512
        pre_ws = zero_block(n_sb + 1, dev)                               # row counts of the pre-refining launch (+ one empty row: idle slots)
        sel_sb = torch.empty((n_u, kk), dtype=torch.int32, device=dev)
        pre_ok = torch.empty((n_u,), dtype=torch.int32, device=dev)
        pre_rows = torch.empty((n_sb * rcap_a,), dtype=torch.int32, device=dev)        # only the listed part is touched
        pre_vals = torch.empty((n_u, kk), dtype=torch.float32, device=dev) if listed else None
        if listed and n_sb >= 32 and lib.trec_get_tuning(b"prerefine_early_dense", 1 if kk <= 16 else 0) != 0:
            # users the int8 bound says nothing about (32 sampled superblocks under tau8) list nothing in the pre-refining launch
            # either: flagged now, as the call after the compaction would.  (The wide route leaves them to that later call: the
            # 64th largest lower bound keeps half of the sampled superblocks of a few users per 100,000 whom the RAISED threshold
            # certifies -- and each flagged user costs an fp32 score slab there.  A user listed here and flagged later is re-done all the same.)
            N.call("trec_topk_dense_users", N.ptr(table), n_sb, n_u, stride, N.ptr(tau), N.ptr(user_err), N.ptr(iop.sb_stats),
                   kpad, CASCADE_DENSE_USER_LIMIT if n_i >= CASCADE_MIN_ITEMS else 30, N.ptr(cands.floor0), N.ptr(cands.flag),
                   N.ptr(cands.n_flagged))
        resident_a = listed and _refine_resident(sb_rows, kpad)
        # (default) the listing launch keeps its maxima by list position and marks the table itself: the threshold kernel behind it
        # reads a few tens of MB back instead of 10 random entries per user of the table (tuning prerefine_marked = 0: the table pass)
        marked = bool(listed) and not resident_a and lib.trec_get_tuning(b"prerefine_marked", 1) != 0
        sel_pos = torch.empty((n_u, kk), dtype=torch.int32, device=dev) if marked else None
        pre_max = torch.empty((n_sb * rcap_a,), dtype=torch.float32, device=dev) if marked else None       # only the listed part is touched
        with _timed("topk_prerefine"):
            if marked:
                N.call("trec_topk_prerefine_rows_pos", N.ptr(sel), N.ptr(sel_max), kk, top_k, sb_per_chunk, n_sb, n_u, N.ptr(uop.src),
                       rcap_a, N.ptr(sel_sb), N.ptr(pre_ws), N.ptr(pre_rows), N.ptr(pre_ok), N.ptr(sel_pos))
            else:
                N.call("trec_topk_prerefine_rows", N.ptr(sel), N.ptr(sel_max), kk, top_k, sb_per_chunk, n_sb, n_u, N.ptr(uop.src),
                       rcap_a, N.ptr(sel_sb), N.ptr(pre_ws), N.ptr(pre_rows), N.ptr(pre_ok))
        if resident_a:
            # the items of a superblock stay in registers, its user list streams through LDS in segments (csrc/refine_resident.hip)
            seg_a, segs_a = _resident_segments(rcap_a)
            n_wgs_a = n_cap_a * kk // seg_a + n_sb + 1
            wg_start_a = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
            wg_map_a = torch.full((n_wgs_a,), n_sb * segs_a, dtype=torch.int32, device=dev)
            N.call("trec_topk_rows_wg_map_ex", N.ptr(pre_ws), n_sb, segs_a, seg_a, N.ptr(wg_start_a), N.ptr(wg_map_a), n_wgs_a)
        elif listed:
            # occupied workgroup slots only: at most pairs / 512 + one partial slot per superblock; the entries the map kernel does
            # not reach point at the empty extra row
            n_wgs_a = n_cap_a * kk // 512 + n_sb + 1
            wg_start_a = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
            wg_map_a = torch.full((n_wgs_a,), n_sb * (rcap_a // 512), dtype=torch.int32, device=dev)
            N.call("trec_topk_rows_wg_map", N.ptr(pre_ws), n_sb, rcap_a // 512, N.ptr(wg_start_a), N.ptr(wg_map_a), n_wgs_a)
        with _timed("score_gemm_blockmax_pre"):
            if resident_a:
                N.call("trec_score_gemm_refine_candidates_resident", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, n_sb, N.ptr(pre_ws), N.ptr(pre_rows), rcap_a, N.ptr(table),
                       stride, N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base),
                       N.ptr(wg_map_a), n_wgs_a, segs_a, seg_a)
            elif marked:
                N.call("trec_score_gemm_refine_candidates_marked", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_sb * rcap_a, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(pre_ws), N.ptr(pre_rows), N.ptr(table), stride,
                       rcap_a // 512, N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base),
                       N.ptr(wg_map_a), n_wgs_a, N.ptr(pre_max))
            elif listed:
                N.call("trec_score_gemm_refine_candidates", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_sb * rcap_a, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(pre_ws), N.ptr(pre_rows), N.ptr(table), stride,
                       rcap_a // 512, N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base),
                       N.ptr(wg_map_a), n_wgs_a)
            else:
                N.call("trec_score_gemm_blockmax_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_sb * rcap_a, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(pre_ws), N.ptr(pre_rows), N.ptr(table), stride,
                       rcap_a // 512)
        with _timed("topk_prerefine"):
            if marked:
                N.call("trec_topk_prerefine_tau_listed", N.ptr(sel_sb), N.ptr(sel_pos), N.ptr(pre_ok), kk, N.ptr(pre_max), rcap_a, n_u,
                       N.ptr(uop.src), N.ptr(uop.stats), N.ptr(user_bias), N.ptr(gstats_all), kpad, N.ptr(tau), N.ptr(pre_vals),
                       N.ptr(cands.floor0))
            else:
                N.call("trec_topk_prerefine_tau", N.ptr(sel_sb), N.ptr(pre_ok), kk, N.ptr(table), stride, n_u, N.ptr(uop.src),
                       N.ptr(uop.stats), N.ptr(user_bias), N.ptr(gstats_all), kpad, N.ptr(tau), listed, N.ptr(pre_vals),
                       N.ptr(cands.floor0))
        if listed:
            cands.pre = (sel_sb, pre_vals)          # (the saved maxima go back into the columns of users re-done from the table)
        if FILTER_DEBUG is not None:
            FILTER_DEBUG.update({"prerefine_users_ok": int(pre_ok.sum().item()), "prerefine_rcap": rcap_a,
                                 "prerefine_row_count_max": int(pre_ws[:n_sb].max().item()), "prerefine_listed": listed,
                                 "prerefine_pairs": int((sel_sb >= 0).sum().item())})
    if cands is None and uop.src is not None:
        tau.masked_fill_(uop.src < 0, float("inf"))     # rows without a source refine nothing (the table-driven tail: A/B reference)
    if one_pass:
        # one pass over the table: a fixed capacity per superblock, slots handed out by atomics (csrc/topk_cascade.hip)
        # (capacities are fractions of the REAL users: a class-sorted layout of few users is mostly rows without a source)
        n_cap = int(uop.n_real or n_u)
        rcap_frac = N.load().trec_get_tuning(b"cascade_rcap_pct", int(100 * CASCADE_ROW_CAPACITY)) / 100.0
        rcap = (int(rcap_frac * n_cap) + 511) // 512 * 512 + 512
        row_user = torch.empty((n_sb * rcap,), dtype=torch.int32, device=dev)      # only the kept pairs' part is touched
        with _timed("topk_rows_compact"):
            N.call("trec_topk_rows_collect", N.ptr(table), n_sb, n_u, stride, N.ptr(tau), N.ptr(user_err),
                   N.ptr(iop.sb_stats), kpad, rcap, N.ptr(row_count), N.ptr(row_user), N.ptr(status))
        if FILTER_DEBUG is not None:
            FILTER_DEBUG.update({"int8_pairs_wanted": int(row_count.sum().item()), "int8_pairs_total": int(n_sb) * int(n_u),
                                 "rcap": rcap, "row_count_max": int(row_count.max().item()),
                                 "hot_superblocks": int((row_count > rcap).sum().item())})
        # "hot" superblocks -- kept by more users than rcap: the few rows that hold a skewed catalogue's most popular items --
        # are refined for EVERY user by a dense launch over that list; the fixed-capacity launch skips them
        hot_cap = max(8, min(N.load().trec_get_tuning(b"cascade_max_hot", CASCADE_MAX_HOT), n_sb))
        hot_list = torch.empty((hot_cap,), dtype=torch.int32, device=dev)
        max_pairs = int(N.load().trec_get_tuning(b"cascade_max_refined_pct", int(100 * CASCADE_MAX_REFINED)) / 100.0 * n_sb * n_cap)
        N.call("trec_topk_rows_hot", N.ptr(row_count), n_sb, rcap, n_u, N.ptr(hot_list), hot_cap, max_pairs, N.ptr(status))
        # Everything that does not need the host's decision is queued BEFORE the host reads the status -- behind the int8 launch,
        # while it still runs: the thresholds above, the users the int8 bound says nothing about, the map of the occupied
        # workgroup slots.  After the read only the refining launches and the finish remain.
        wg_map, wg_cap = None, 0
        if cands is not None:
            if n_sb >= 32:                              # users the int8 bound says nothing about are flagged now, not listed for
                N.call("trec_topk_dense_users", N.ptr(table), n_sb, n_u, stride, N.ptr(tau), N.ptr(user_err), N.ptr(iop.sb_stats),
                       kpad, CASCADE_DENSE_USER_LIMIT if n_i >= CASCADE_MIN_ITEMS else 30, N.ptr(cands.floor0), N.ptr(cands.flag),
                       N.ptr(cands.n_flagged))             # (small catalogues: the k-th largest of few maxima keeps half the rows of anybody)
            resident = _refine_resident(sb_rows, kpad)
            if resident:
                # the items of a superblock resident, its user list in segments (csrc/refine_resident.hip): the map lists the
                # occupied (superblock, segment) slots; the rest of its entries stay idle
                seg_r, segs_r = _resident_segments(rcap)
                wg_cap = min(n_sb * segs_r, max_pairs // seg_r + n_sb + 1)
                wg_start = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
                wg_map = torch.full((wg_cap,), n_sb * segs_r, dtype=torch.int32, device=dev)
                N.call("trec_topk_rows_wg_map_ex", N.ptr(row_count), n_sb, segs_r, seg_r, N.ptr(wg_start), N.ptr(wg_map), wg_cap)
            elif N.load().trec_get_tuning(b"cascade_wg_map", 1) != 0:
                # only the workgroup slots that hold rows are launched (98k of the 1.9M of the [n_sb][rcap / 512] grid at 1M x 1M)
                wg_cap = min(n_sb * (rcap // 512), max_pairs // 512 + n_sb + 1)
                wg_start = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
                wg_map = torch.empty((wg_cap,), dtype=torch.int32, device=dev)
                N.call("trec_topk_rows_wg_map", N.ptr(row_count), n_sb, rcap // 512, N.ptr(wg_start), N.ptr(wg_map), wg_cap)
        # the overflow flag is read HERE, before the bf16 launches: when the int8 bound is too loose, refining
        # close to half of all pairs and then running the filter's tail on the result would only be thrown away.  One host
        # read per call; item shards agree on it (MAX).
        rows, overflow, n_hot = status.tolist()
        if stats_exchange is not None:
            overflow = float(stats_exchange(torch.tensor([float(overflow)], device=dev)).item())
        if overflow:
            return None, stride, (int(rows), True), tau, None
        n_hot = min(int(n_hot), hot_cap)                # the hot launch's grid: nothing is launched when no superblock is hot
        n_wgs = 0
        if wg_map is not None and cands is not None and resident:
            n_wgs = min(wg_cap, (int(rows) - n_hot * ((n_u + 511) // 512 * 512)) // seg_r + n_sb + 1)
        elif wg_map is not None:
            n_wgs = min(wg_cap, (int(rows) - n_hot * ((n_u + 511) // 512 * 512)) // 512)
        # (user batches in a pipeline: from here on the launches go to the tail stream, next to the following batch's int8 stage)
        tail = _tail_of(tail_stream if cands is not None else None, table, row_count, row_user, hot_list, wg_map,
                        *((cands.floor0, cands.n, cands.items) if cands is not None else ()))
        with tail, _timed("score_gemm_blockmax_grouped"):
            if cands is not None and resident:
                N.call("trec_score_gemm_refine_candidates_resident", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, n_sb, N.ptr(row_count), N.ptr(row_user), rcap, N.ptr(table),
                       stride, N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base),
                       N.ptr(wg_map), n_wgs, segs_r, seg_r)
            elif cands is not None:
                N.call("trec_score_gemm_refine_candidates", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_sb * rcap, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(row_count), N.ptr(row_user), N.ptr(table), stride,
                       rcap // 512, N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base),
                       N.ptr(wg_map), n_wgs)
            else:
                N.call("trec_score_gemm_blockmax_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_sb * rcap, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(row_count), N.ptr(row_user), N.ptr(table), stride,
                       rcap // 512)
        with _tail_of(tail_stream if cands is not None else None), _timed("score_gemm_blockmax_hot"):
            if n_hot == 0:
                pass
            elif cands is not None:
                N.call("trec_score_gemm_refine_candidates_hot", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_u, n_i,
                       N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(hot_list), n_hot, N.ptr(table), stride,
                       N.ptr(cands.floor0), N.ptr(cands.n), N.ptr(cands.items), cands.cap, int(item_index_base))
            else:
                N.call("trec_score_gemm_blockmax_hot", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, n_u, n_i, N.ptr(user_bias),
                       N.ptr(item_bias), sb_rows, N.ptr(hot_list), n_hot, N.ptr(table), stride)
        return table, stride, (int(rows), False), tau, cands
    n_ublk = N.query("trec_topk_rows_user_blocks", n_u)
    block_off = torch.empty((n_sb * n_ublk,), dtype=torch.int32, device=dev)
    row_total = torch.empty((n_sb,), dtype=torch.int32, device=dev)
    row_pad = torch.empty((n_sb,), dtype=torch.int32, device=dev)
    pstart = torch.empty((n_sb + 1,), dtype=torch.int64, device=dev)
    cap_rows = (int(CASCADE_MAX_REFINED * n_sb * n_u) + 511) // 512 * 512 + 512 * n_sb
    row_user = torch.empty((cap_rows,), dtype=torch.int32, device=dev)          # only the kept pairs' part is touched
    rblock_chunk = torch.empty((cap_rows // 512,), dtype=torch.int32, device=dev)
    with _timed("topk_rows_compact"):
        N.call("trec_topk_rows_count", N.ptr(table), n_sb, n_u, stride, N.ptr(tau), N.ptr(user_err), N.ptr(iop.sb_stats),
               kpad, N.ptr(block_off), N.ptr(row_total), N.ptr(row_pad), N.ptr(pstart), cap_rows, N.ptr(status))
        N.call("trec_topk_rows_fill", N.ptr(table), n_sb, n_u, stride, N.ptr(tau), N.ptr(user_err), N.ptr(iop.sb_stats),
               kpad, N.ptr(block_off), N.ptr(row_total), N.ptr(pstart), cap_rows, N.ptr(status), N.ptr(row_user),
               N.ptr(rblock_chunk))
    with _timed("score_gemm_blockmax_grouped"):
        N.call("trec_score_gemm_blockmax_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), kpad, cap_rows, n_i,
               N.ptr(user_bias), N.ptr(item_bias), sb_rows, N.ptr(rblock_chunk), N.ptr(row_user), N.ptr(table), stride, 0)
    rows, overflow = status.tolist()[:2]                # (the two-pass form: its fill pass and grouped launch idle after an overflow)
    if stats_exchange is not None:
        overflow = float(stats_exchange(torch.tensor([float(overflow)], device=dev)).item())
    return (None if overflow else table), stride, (int(rows), bool(overflow)), tau, None


def _filter_tail(uop, iop, blockmax, bm_stride, n_u, n_sb, k, user_bias, item_bias, item_index_base, sb_rows, variant, ksel,
                 cap, floor, flag, n_flagged, rows_wg, wide=False, keys_count=None):
    """Stages 2b-4 of the filtered top-k on a table of superblock maxima: every superblock reaching the user's floor
    (at most ``ksel``), grouped bf16 re-scoring with ``cap``-entry lists, exact fp32 finish.  ``wide``: the second pass over
    the users the first one flagged -- the finish kernel without capacity limits."""
    dev = uop.bf16.device
    n_i, kpad = iop.n, uop.kpad
    n_pairs = n_u * ksel
    if keys_count is not None:                          # the one-pass scan already produced them (trec_topk_prune_candidates)
        keys, count = keys_count
    else:
        keys = torch.empty((n_pairs,), dtype=torch.int32, device=dev)
        count = torch.empty((n_u,), dtype=torch.int32, device=dev)
        with _timed("topk_collect_blocks"):
            N.call("trec_topk_collect_blocks", N.ptr(blockmax), n_sb, n_u, bm_stride, N.ptr(floor), ksel, N.ptr(keys),
                   N.ptr(count), N.ptr(flag), N.ptr(n_flagged))
    if FILTER_DEBUG is not None and not wide:
        FILTER_DEBUG["flagged_after_collect"] = int(n_flagged.item())
        _debug_counts("kept_superblocks", count)
        true_cnt = torch.zeros((n_u,), dtype=torch.int32, device=dev)       # without the ksel cap
        for s0 in range(0, n_sb, 64):
            true_cnt += (blockmax[s0:s0 + 64, :n_u] >= floor[None, :]).sum(0, dtype=torch.int32)
        _debug_counts("kept_superblocks_uncapped", true_cnt)
    # ---- stage 3a: group the kept (user, slot) pairs by superblock, pad groups to whole workgroups, gather bf16 rows
    indptr_t, users_t, perm_t = group_pairs_by_item(None, keys, ksel, n_sb + 1)
    cnt_pad = torch.empty((n_sb + 1,), dtype=torch.int32, device=dev)
    N.call("trec_topk_pad_counts", N.ptr(indptr_t), n_sb, rows_wg, N.ptr(cnt_pad))
    pstart = torch.empty((n_sb + 2,), dtype=torch.int64, device=dev)
    ws64 = torch.empty(((n_sb + 1 + 1023) // 1024 + 1,), dtype=torch.int64, device=dev)
    N.call("trec_exclusive_scan_i32", N.ptr(cnt_pad), n_sb + 1, N.ptr(ws64), N.ptr(pstart))
    max_rows = (n_pairs + min(n_sb, n_pairs) * (rows_wg - 1) + rows_wg - 1) // rows_wg * rows_wg
    row_user = torch.empty((max_rows,), dtype=torch.int32, device=dev)
    row_pair = torch.empty((max_rows,), dtype=torch.int32, device=dev)
    rblock_chunk = torch.empty((max_rows // rows_wg,), dtype=torch.int32, device=dev)
    with _timed("topk_fill_groups"):
        N.call("trec_topk_fill_groups_index", N.ptr(pstart), N.ptr(indptr_t), N.ptr(users_t), N.ptr(perm_t), n_sb, rows_wg,
               max_rows, N.ptr(row_user), N.ptr(row_pair), N.ptr(rblock_chunk))
    # ---- stage 3b: bf16 re-scoring of the kept superblocks, every item >= floor listed (independent lists: bit 4);
    # the user rows / biases / floors are fetched through row_user, only item ids are written
    pi = torch.empty((n_pairs * 2, cap), dtype=torch.int32, device=dev)        # only the kept pairs' lists are touched
    with _timed("score_gemm_topk_grouped"):
        N.call("trec_score_gemm_topk_grouped", N.ptr(uop.bf16), N.ptr(iop.bf16), DTYPE_BF16, kpad, max_rows, n_i,
               item_index_base, N.ptr(user_bias), N.ptr(item_bias), MODE_DOT, None, None, sb_rows, N.ptr(rblock_chunk),
               N.ptr(row_pair), N.ptr(floor), cap, None, N.ptr(pi), (variant & 1) | 16, N.ptr(row_user))
    # ---- stage 4: exact fp32 scores of the survivors, exact top-k
    ov = torch.empty((n_u, int(k)), dtype=torch.float32, device=dev)
    oi = torch.empty((n_u, int(k)), dtype=torch.int32, device=dev)
    with _timed("topk_filter_finish"):
        N.call("trec_topk_filter_finish_wide" if wide else "trec_topk_filter_finish", N.ptr(pi), cap, ksel, N.ptr(count),
               N.ptr(uop.f32), N.ptr(iop.f32), kpad, kpad, uop.d, N.ptr(user_bias), N.ptr(item_bias), item_index_base, n_u,
               int(k), N.ptr(ov), N.ptr(oi), N.ptr(flag), N.ptr(n_flagged))
    return ov, oi, count


WIDE_PASS_BYTES = 4 << 30     # list workspace of one launch chain of the wide pass (bounds the flagged users per chain)


def _wide_second_pass(uop, iop, blockmax, bad, n_sb, k, user_bias, item_bias, item_index_base, sb_rows, variant, floor, rows_wg,
                      ksel_w, gstats=None):
    """The users the first pass could not certify (more than FILTER_KSEL kept superblocks, more than 64 survivors, a full
    8-entry list) again, on THEIR columns of the table that already exists: ``ksel_w`` slots, 16-entry lists, a finish without
    survivor limit.  Returns (values, ids, flag) for the rows ``bad``; what is still flagged goes to the next tier.
    ``floor`` None (the first pass went through candidate lists and never derived the table's floor): it is derived here, from
    the columns of these users -- on an item shard from the LOCAL table, which is sound (k local entries certify k items) if
    less selective than the exchanged one."""
    dev = uop.bf16.device
    per_chain = max(1024, WIDE_PASS_BYTES // (ksel_w * 2 * 16 * 4))
    out_v, out_i, out_f = [], [], []
    with _timed("topk_filter_wide_pass"):
        for s0 in range(0, int(bad.numel()), per_chain):
            b = bad[s0:s0 + per_chain]
            n_b = int(b.numel())
            sub = FilterOperand()
            sub.n, sub.d, sub.kpad = n_b, uop.d, uop.kpad
            sub.bf16 = uop.bf16.index_select(0, b)
            sub.f32 = uop.f32.index_select(0, b)
            table_b = blockmax.index_select(1, b)
            ub = user_bias.index_select(0, b) if user_bias is not None else None
            flag_b = torch.zeros((n_b,), dtype=torch.int32, device=dev)
            n_flagged_b = torch.zeros((1,), dtype=torch.int32, device=dev)
            if floor is None:
                sub.stats = uop.stats.index_select(0, b)
                sel_b = torch.empty((n_b, int(k)), dtype=torch.int32, device=dev)
                tau_b = torch.empty((n_b,), dtype=torch.float32, device=dev)
                floor_b = torch.empty((n_b,), dtype=torch.float32, device=dev)
                N.call("trec_topk_select_blocks", N.ptr(table_b), n_sb, n_b, n_b, int(k), N.ptr(sel_b), None, N.ptr(tau_b))
                N.call("trec_topk_filter_floor", N.ptr(tau_b), N.ptr(sub.stats), N.ptr(ub), N.ptr(gstats), uop.kpad, n_b,
                       N.ptr(floor_b), N.ptr(flag_b), N.ptr(n_flagged_b))
            else:
                floor_b = floor.index_select(0, b)
            wv, wi, _ = _filter_tail(sub, iop, table_b, n_b, n_b, n_sb, k, ub, item_bias, item_index_base, sb_rows, variant,
                                     ksel_w, 16, floor_b, flag_b, n_flagged_b, rows_wg, wide=True)
            out_v.append(wv); out_i.append(wi); out_f.append(flag_b)
            del table_b, sub
    return torch.cat(out_v), torch.cat(out_i), torch.cat(out_f)


def score_topk_filtered(uop, iop, k, user_bias=None, item_bias=None, item_index_base=0, sb_rows=None, variant=1,
                        n_chunks=None, floor_exchange=None, stats_exchange=None, ksel=None, prefilter=None, finish_lanes=0):
    """See _score_topk_filtered.  A user operand sorted by int8 scale class (``uop.src``) is handled here: the user biases
    follow the operand's row order on the way in (``uop.bias_sorted`` when this ``user_bias`` was given to the preparation),
    the results leave in the caller's order (the finish kernel writes through ``uop.src``: no permutation pass).  Item shards:
    the per-user exchanges then carry users in the operand's order -- the same on every rank, because the user side is
    replicated and the sort is deterministic.  ``finish_lanes=16`` (item shards of a run over >= 4 ranks: a user lists ~27 / N
    candidates per shard): the exact finish packs four users into a wave; a user with more than 16 candidates is flagged and
    re-done on its table column like any other flagged user."""
    if uop.src is None:
        return _score_topk_filtered(uop, iop, k, user_bias, item_bias, item_index_base, sb_rows, variant, n_chunks,
                                    floor_exchange, stats_exchange, ksel, prefilter, finish_lanes=finish_lanes)
    if user_bias is None:
        ub = None
    elif uop.bias_sorted is not None and uop.bias_ref is user_bias:
        ub = uop.bias_sorted
    else:
        ub = user_bias.reshape(-1).index_select(0, uop.perm).masked_fill_(uop.pad, 0.0)
    n_batches = cascade_user_batches(uop, iop, prefilter, floor_exchange, stats_exchange)
    if n_batches > 1:
        sv, si = _score_topk_filtered_pipelined(uop, iop, k, ub, item_bias, item_index_base, sb_rows, variant, n_chunks, ksel,
                                                n_batches)
        pos = uop.pos.long()
        ov, oi = sv.index_select(0, pos), si.index_select(0, pos)
    else:
        ov, oi = _score_topk_filtered(uop, iop, k, ub, item_bias, item_index_base, sb_rows, variant, n_chunks,
                                      floor_exchange, stats_exchange, ksel, prefilter, caller_order=True, finish_lanes=finish_lanes)
    LAST_FILTER_STATS["users"] = int(uop.n_real)
    LAST_FILTER_STATS["layout_rows"] = int(uop.n)
    return ov, oi


CASCADE_PIPELINE_MIN_ROWS = 131072     # user rows per batch below which the two-stream pipeline is not worth its launches
_TAIL_STREAMS = {}


def cascade_user_batches(uop, iop, prefilter, floor_exchange, stats_exchange):
    """User batches of the two-stream pipeline (tuning ``cascade_user_batches``; default 1 = off): the int8 stage of batch
    b + 1 runs next to the bf16 refinement and the finish of batch b -- the first is bound by the matrix pipe at the power limit,
    the others by gathers and latencies.  Single process, class-sorted users, a catalogue the cascade is used on.
    MEASURED at 1M x 1M: 4 batches 98.6-98.8 ms against 95.7-96.5 for one -- the tail kernels do run next to the
    int8 launch (their event time stretches to its ~20 ms) but the chip is at its power limit either way: the int8 launches
    lose what the hidden tail gains, and four smaller launches of everything cost more than one.  Kept as a knob."""
    want = N.load().trec_get_tuning(b"cascade_user_batches", 1)
    if want <= 1 or prefilter != "int8" or floor_exchange is not None or stats_exchange is not None or uop.wg_rows is None:
        return 1
    if uop.kpad not in (64, 128) or iop.cascade_too_loose or not cascade_lists_candidates():
        return 1
    return max(1, min(int(want), int(uop.n_real or uop.n) // CASCADE_PIPELINE_MIN_ROWS))


def _rows_of(uop, r0, r1):
    """Rows r0 .. r1 (multiples of the int8 workgroup height) of a class-sorted user operand, as views."""
    sub = FilterOperand()
    sub.n, sub.n_real, sub.d, sub.kpad = r1 - r0, r1 - r0, uop.d, uop.kpad
    sub.bf16, sub.f32, sub.stats = uop.bf16[r0:r1], uop.f32[r0:r1], uop.stats[r0:r1]
    sub.i8, sub.stats8 = uop.i8[r0:r1], uop.stats8[r0:r1]
    sub.src = uop.src[r0:r1]
    sub.wg_rows = uop.wg_rows
    sub.wg_scale = uop.wg_scale[r0 // uop.wg_rows:r1 // uop.wg_rows]
    sub.wg_class = uop.wg_class[r0 // uop.wg_rows:r1 // uop.wg_rows]
    sub.gmax, sub.ladder, sub.class_used = uop.gmax, uop.ladder, uop.class_used
    return sub


def _score_topk_filtered_pipelined(uop, iop, k, user_bias, item_bias, item_index_base, sb_rows, variant, n_chunks, ksel, n_batches):
    """_score_topk_filtered(prefilter="int8") over ``n_batches`` row ranges of the class-sorted operand, two streams: everything
    up to the host's read of the compaction status runs on the caller's stream, the bf16 refinement and the finish of a batch on
    a second one -- next to the following batch's int8 stage.  The flagged users of all batches are re-done at the end."""
    sb_rows = int(sb_rows or SUPERBLOCK_ROWS)
    top_k = 10 if int(k) <= 10 else 16
    if uop.i8 is None or iop.i8 is None or iop.sb_rows != sb_rows or iop.bias_owner is None or iop.bias_owner() is not uop:
        score_prep_i8_pair(uop, iop, item_bias, sb_rows, top_k)          # ONCE for all batches (the classes in use, the item biases)
    dev = uop.bf16.device
    main = torch.cuda.current_stream()
    tail = _TAIL_STREAMS.get(dev)
    if tail is None:
        # high priority: a tail kernel's workgroups take the slots the int8 kernel's retiring workgroups free (at equal priority
        # the int8 launch -- queued first, 16,000+ workgroups -- keeps every slot and the tail only runs once it has drained)
        prio = -1 if N.load().trec_get_tuning(b"cascade_tail_priority", 1) != 0 else 0
        tail = _TAIL_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=prio)
    wgs = int(uop.n) // int(uop.wg_rows)
    bounds = [int(uop.wg_rows) * (wgs * b // n_batches) for b in range(n_batches + 1)]
    parts = []
    for b in range(n_batches):
        r0, r1 = bounds[b], bounds[b + 1]
        sub = _rows_of(uop, r0, r1)
        iop.bias_owner = __import__("weakref").ref(sub)         # (the tables made for the whole operand serve its row ranges)
        ub = user_bias[r0:r1] if user_bias is not None else None
        parts.append(_score_topk_filtered(sub, iop, k, ub, item_bias, item_index_base, sb_rows, variant, n_chunks, None, None,
                                          ksel, "int8", tail_stream=tail))
        if not isinstance(parts[-1], _Pending):
            parts[-1] = (parts[-1], dict(LAST_FILTER_STATS))
    main.wait_stream(tail)
    out_v, out_i, stats = [], [], []
    for part in parts:
        if isinstance(part, _Pending):
            v, i = part.complete()
            part = ((v, i), dict(LAST_FILTER_STATS))
        out_v.append(part[0][0]); out_i.append(part[0][1]); stats.append(part[1])
    LAST_FILTER_STATS.clear()
    LAST_FILTER_STATS.update(stats[0])
    for key in ("refined_rows", "users", "flagged_users", "flagged_after_wide_pass", "flagged_after_wide_pass_2",
                "users_on_fp32_fallback"):
        if any(key in st for st in stats):
            LAST_FILTER_STATS[key] = sum(st.get(key, 0) for st in stats)
    if all("candidates_per_user" in st for st in stats):
        LAST_FILTER_STATS["candidates_per_user"] = sum(st["candidates_per_user"] * st["users"] for st in stats) / max(1, uop.n)
    if any(st.get("prefilter") != "int8" for st in stats):
        LAST_FILTER_STATS["prefilter"] = "; ".join(sorted(set(str(st.get("prefilter")) for st in stats)))
    LAST_FILTER_STATS["user_batches"] = n_batches
    return torch.cat(out_v), torch.cat(out_i)


def _score_topk_filtered(uop, iop, k, user_bias=None, item_bias=None, item_index_base=0, sb_rows=None, variant=1,
                         n_chunks=None, floor_exchange=None, stats_exchange=None, ksel=None, prefilter=None, tail_stream=None,
                         caller_order=False, finish_lanes=0):
    """EXACT fp32 top-k (values and ids bit-identical to ``score_topk(..., DTYPE_F32)`` and to the oracle) with the
    score matrix contracted ONCE on bf16 MFMA: the bf16 stage-1 maxima and the bf16 re-scoring act as a filter with a
    proven error bound (csrc/topk_filter.hip), the survivors (~15 items per user at 1M x 1M) are re-scored by the
    reference's k-ordered fp32 chain.  ``uop`` / ``iop``: FilterOperand (iop with gstats).  Dot / cosine scores.
    Item shards: ``floor_exchange`` as in score_topk_two_stage, ``stats_exchange(gstats) -> gstats`` = all-reduce MAX of
    the item-side maxima (the bound must cover every shard's items).  Users the filter cannot certify (its capacity
    limits, non-finite bounds) are re-done on the exact fp32 MFMA path; their number is in LAST_FILTER_STATS.
    ``prefilter="int8"``: stage 1 becomes the cascade of csrc/topk_cascade.hip -- an exact-integer int8 MFMA pass over
    everything, the bf16 kernel only on the (superblock, user) pairs the int8 bound cannot rule out (kpad 64 / 128).
    ``caller_order`` (a class-sorted ``uop``): the lists come back as [uop.n_real, k] in the CALLER's row order -- the candidate
    finish writes them there through ``uop.src``; the other tails permute at the end -- instead of [uop.n, k] in layout order."""
    if not 1 <= int(k) <= 16:
        raise ValueError("fused top-k supports k <= 16 (got %d)" % k)
    cap = 8              # stage-3 lists hold survivors of ONE (user, superblock, half-wave): 0-2 typically; full -> exact fallback
    dev = uop.bf16.device
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad
    if iop.kpad != kpad or iop.gstats is None:
        raise ValueError("score_topk_filtered: operands must share kpad and the item side needs gstats")
    sb_rows = int(sb_rows or SUPERBLOCK_ROWS)
    n_sb = (n_i + sb_rows - 1) // sb_rows
    if n_sb < int(k):
        raise ValueError("score_topk_filtered needs at least k superblocks of items")
    ksel = min(int(ksel or FILTER_KSEL), n_sb)
    ksel = max(ksel, int(k))
    rows_wg = N.query("trec_score_rows_per_workgroup", DTYPE_BF16, kpad)
    if n_chunks is None:
        rblocks = (n_u + rows_wg - 1) // rows_wg
        # (few users: at most CASCADE_MAX_CHUNKS item chunks -- the selection walks chunks x k list entries per user)
        n_chunks = max(1, min(n_sb, N.load().trec_get_tuning(b"cascade_max_chunks", CASCADE_MAX_CHUNKS), -(-32 * 768 // rblocks)))
    LAST_FILTER_STATS.clear()
    blockmax, bm_stride, cascade_status, tau8, cands = None, n_u, None, None, None
    # item shards: the item-side maxima behind both bounds (norms, rounding-error norms, |bias|) are MAX-reduced ONCE per call
    gstats = iop.gstats
    if stats_exchange is not None:
        gstats = stats_exchange(gstats).contiguous()
    if prefilter == "int8" and kpad in (64, 128) and sb_rows % 128 == 0 and not (iop.cascade_too_loose and
                                                                                 floor_exchange is None):
        # ---- stages 0 + 1: int8 maxima everywhere, bf16 maxima where a top-k item can be
        blockmax, bm_stride, cascade_status, tau8, cands = _cascade_stage1(uop, iop, k, user_bias, item_bias, sb_rows, n_sb,
                                                                           n_chunks, floor_exchange, stats_exchange, gstats,
                                                                           item_index_base, tail_stream)
        rows, overflow = cascade_status
        if overflow:
            # the int8 bound was too loose for this data: nothing was refined.  bf16 does stage 1; the next user batches
            # against this catalogue skip the attempt (item shards keep trying: every rank must take the same path and the
            # flag is local)
            r = _score_topk_filtered(uop, iop, k, user_bias, item_bias, item_index_base, sb_rows, variant, n_chunks,
                                     floor_exchange, stats_exchange, ksel, None, caller_order=caller_order, finish_lanes=finish_lanes)
            LAST_FILTER_STATS["prefilter"] = "int8 (too loose: bf16 stage 1 instead)"
            LAST_FILTER_STATS["refined_rows"] = int(rows)
            iop.cascade_too_loose = True
            return r
        cascade_rows = int(rows)
    elif prefilter not in (None, "int8"):
        raise ValueError("unknown prefilter %r" % (prefilter,))
    if blockmax is None:
        # ---- stage 1: bf16 superblock maxima
        bm_stride = n_u
        blockmax = torch.empty((n_sb, n_u), dtype=torch.float32, device=dev)
        with _timed("score_gemm_blockmax"):
            N.call("trec_score_gemm_blockmax", N.ptr(uop.bf16), N.ptr(iop.bf16), DTYPE_BF16, kpad, n_u, n_i,
                   N.ptr(user_bias), N.ptr(item_bias), MODE_DOT, None, None, sb_rows, n_chunks, N.ptr(blockmax), n_u,
                   variant | 32)          # bit 5: filter use -> the 16x16x32 MFMA form (any summation order obeys the bound)
    # ---- stage 2: tau = k-th largest superblock maximum (a floor of the k-th best bf16 score), floor = tau - 2 eps (proven
    # bound, csrc/topk_filter.hip), then every superblock reaching the floor.  Two passes over the table -- or, behind the int8
    # stage, ONE: tau8 gives a provisional floor (tau >= tau8 - eps, so the final floor is >= tau8 - 3 eps), the scan keeps the
    # k largest entries AND lists the entries above it, and the final floor only prunes those ~45 candidates per user
    kk = int(k)
    if cands is not None:
        # ---- the refining launches listed every item that can still reach the top-k: finish from the lists (stages 2-3 gone)
        # (class-sorted users: user u's lists are written to the CALLER's row uop.src[u]; rows without one write nothing)
        out_index = uop.src if (caller_order and uop.src is not None) else None
        n_out = int(uop.n_real) if out_index is not None else n_u
        ov = torch.empty((n_out, kk), dtype=torch.float32, device=dev)
        oi = torch.empty((n_out, kk), dtype=torch.int32, device=dev)
        flag, n_flagged = cands.flag, cands.n_flagged
        # single process, no lane count asked for: lists are ~15 entries long (a few users hold hundreds) -- 16 lanes x 4 candidates
        # per user, four users per wave, and the wave-per-user form only for the users beyond 64 entries (tuning finish_mixed: 0 =
        # a wave per user for everybody, the A/B reference; 1 / 2 / 4 = candidates per lane)
        mixed = int(N.load().trec_get_tuning(b"finish_mixed", 4)) if not finish_lanes and cands.cap <= 256 else 0
        with _tail_of(tail_stream, ov, oi, flag, n_flagged, gstats), _timed("topk_filter_finish"):
            if mixed in (1, 2, 4):
                over_count = zero_block(2, dev)             # (allocated inside the tail-stream context: they live on the stream they are used on)
                over_list = torch.empty((n_u,), dtype=torch.int32, device=dev)
                N.call("trec_topk_candidates_finish_mixed", N.ptr(cands.n), N.ptr(cands.items), cands.cap, N.ptr(cands.floor0),
                       N.ptr(uop.stats), N.ptr(gstats), N.ptr(uop.f32), N.ptr(iop.f32), kpad, kpad, uop.d, N.ptr(user_bias),
                       N.ptr(item_bias), item_index_base, n_u, kk, N.ptr(ov), N.ptr(oi), N.ptr(flag), N.ptr(n_flagged),
                       N.ptr(out_index), mixed, N.ptr(over_list), N.ptr(over_count))
            else:
                N.call("trec_topk_candidates_finish", N.ptr(cands.n), N.ptr(cands.items), cands.cap, N.ptr(cands.floor0),
                       N.ptr(uop.stats), N.ptr(gstats), N.ptr(uop.f32), N.ptr(iop.f32), kpad, kpad, uop.d, N.ptr(user_bias),
                       N.ptr(item_bias), item_index_base, n_u, kk, N.ptr(ov), N.ptr(oi), N.ptr(flag), N.ptr(n_flagged),
                       N.ptr(out_index), int(finish_lanes or 0))

        def complete(cands=cands, blockmax=blockmax):
            if FILTER_DEBUG is not None:
                _debug_counts("candidates", cands.n)
            # ONE host read: the flagged-user counter (queued behind the finish kernel)
            n_bad = int(n_flagged.item())
            if n_bad and cands.pre is not None:
                # the pre-refined pairs of the users to re-do: their bf16 maxima return to the table (the compaction saw -inf there)
                sel_sb, pre_vals = cands.pre
                bad_u = torch.nonzero(flag, as_tuple=False).reshape(-1)
                sb_b = sel_sb.index_select(0, bad_u).long()
                hit = sb_b >= 0
                blockmax[sb_b[hit], bad_u.reshape(-1, 1).expand_as(sb_b)[hit]] = pre_vals.index_select(0, bad_u)[hit]
            LAST_FILTER_STATS.clear()
            LAST_FILTER_STATS.update({"prefilter": "int8", "refined_rows": cascade_rows, "users": n_u, "flagged_users": n_bad,
                                      "tail": "candidate lists", "candidates_cap": cands.cap})
            if cands.pre is not None:
                # (at most: a slot whose superblock list was full stays with the compaction)
                LAST_FILTER_STATS["prerefined_pairs"] = int(uop.n_real or n_u) * int(k)
            if CANDIDATE_STATS:                           # diagnostics (a reduction over the counters + a host read): off the product path
                LAST_FILTER_STATS["candidates_per_user"] = float(cands.n.clamp(max=cands.cap).sum().item()) / \
                    max(1, int(uop.n_real or n_u))        # (per real user: the layout's rows without a source list nothing)
            return _redo_flagged(uop, iop, blockmax, flag, n_bad, n_sb, k, user_bias, item_bias, item_index_base, sb_rows,
                                 variant, None, rows_wg, ksel, gstats, ov, oi, out_index)
        cands = blockmax = None
        if tail_stream is not None:
            return _Pending(complete)
        return complete()
    tau = torch.empty((n_u,), dtype=torch.float32, device=dev)
    sel_max = torch.empty((kk, n_u), dtype=torch.float32, device=dev) if floor_exchange is not None else None
    floor = torch.empty((n_u,), dtype=torch.float32, device=dev)
    flag = torch.empty((n_u,), dtype=torch.int32, device=dev)
    n_flagged = torch.zeros((1,), dtype=torch.int32, device=dev)
    keys_count = None
    # (it pays on big tables: 4.0 -> 3.55 ms of table passes at 1M users x 1,954 superblocks; on a 32,768-user batch the two
    # passes cost 0.15 ms and the extra launches more than they save)
    # tuning filter_scan_one_pass: 0 = never, 1 = by table size (default), 2 = always (tests)
    mode1p = N.load().trec_get_tuning(b"filter_scan_one_pass", 1)
    one_pass = tau8 is not None and cascade_status is not None and mode1p != 0 and \
        (mode1p == 2 or n_u * n_sb >= FILTER_ONE_PASS_MIN_ENTRIES)
    if one_pass:
        cand_cap = FILTER_CANDIDATES
        floor0 = torch.empty((n_u,), dtype=torch.float32, device=dev)
        N.call("trec_topk_filter_floor_ex", N.ptr(tau8), N.ptr(uop.stats), N.ptr(user_bias), N.ptr(gstats), kpad, n_u, 4.0,
               N.ptr(floor0), None, None)
        cand_s = torch.empty((cand_cap, n_u), dtype=torch.int32, device=dev)       # slot-major; only the first ~45 rows are touched
        cand_v = torch.empty((cand_cap, n_u), dtype=torch.float32, device=dev)
        cand_n = torch.empty((n_u,), dtype=torch.int32, device=dev)
        with _timed("topk_select_blocks"):
            N.call("trec_topk_scan_blocks", N.ptr(blockmax), n_sb, n_u, bm_stride, kk, N.ptr(floor0), cand_cap, N.ptr(sel_max),
                   N.ptr(tau), N.ptr(cand_s), N.ptr(cand_v), N.ptr(cand_n))
    else:
        sel = torch.empty((n_u, kk), dtype=torch.int32, device=dev)
        with _timed("topk_select_blocks"):
            N.call("trec_topk_select_blocks", N.ptr(blockmax), n_sb, n_u, bm_stride, kk, N.ptr(sel), N.ptr(sel_max),
                   N.ptr(tau))
    if floor_exchange is not None:                  # item shards: the k-th largest maximum over ALL shards
        tau = floor_exchange(sel_max).contiguous()
    N.call("trec_topk_filter_floor", N.ptr(tau), N.ptr(uop.stats), N.ptr(user_bias), N.ptr(gstats), kpad, n_u,
           N.ptr(floor), N.ptr(flag), N.ptr(n_flagged))
    if uop.src is not None:
        floor.masked_fill_(uop.src < 0, float("inf"))   # rows without a source keep nothing: no lists, no survivors, never flagged
    if FILTER_DEBUG is not None:
        FILTER_DEBUG["flagged_after_floor"] = int(n_flagged.item())
    if one_pass:
        keys = torch.empty((n_u * ksel,), dtype=torch.int32, device=dev)
        count = torch.empty((n_u,), dtype=torch.int32, device=dev)
        redo = torch.empty((n_u,), dtype=torch.int32, device=dev)
        with _timed("topk_collect_blocks"):
            N.call("trec_topk_prune_candidates", N.ptr(cand_s), N.ptr(cand_v), N.ptr(cand_n), cand_cap, N.ptr(floor), ksel, n_u,
                   N.ptr(keys), N.ptr(count), N.ptr(flag), N.ptr(n_flagged), N.ptr(redo))
            # users with more entries above the provisional floor than the candidate list holds (a loose int8 bound): their
            # column is collected from the table after all; workgroups of 256 users without such a user exit at once
            N.call("trec_topk_collect_blocks_masked", N.ptr(blockmax), n_sb, n_u, bm_stride, N.ptr(floor), ksel, N.ptr(keys),
                   N.ptr(count), N.ptr(flag), N.ptr(n_flagged), N.ptr(redo))
        keys_count = (keys, count)
        del cand_s, cand_v
    ov, oi, count = _filter_tail(uop, iop, blockmax, bm_stride, n_u, n_sb, k, user_bias, item_bias, item_index_base, sb_rows,
                                 variant, ksel, cap, floor, flag, n_flagged, rows_wg, wide=False, keys_count=keys_count)
    # ---- users the filter could not certify (one host read of a counter): a wide second pass, then the exact fp32 MFMA path
    if uop.src is not None:
        # (this table-driven tail is the A/B reference of a class-sorted operand: the columns of layout rows without a source --
        # idle int8 workgroups never wrote them -- may hold anything, NaN included, and get flagged; they have no result)
        flag.masked_fill_(uop.src < 0, 0)
        n_flagged = flag.sum().reshape(1)
    n_bad = int(n_flagged.item())
    if cascade_status is not None:
        LAST_FILTER_STATS["prefilter"] = "int8"
        LAST_FILTER_STATS["refined_rows"] = cascade_rows
    LAST_FILTER_STATS.update({"users": n_u, "flagged_users": n_bad, "ksel": ksel,
                              "kept_superblocks_per_user": float(count.sum().item()) / max(1, n_u)})
    ov, oi = _redo_flagged(uop, iop, blockmax, flag, n_bad, n_sb, k, user_bias, item_bias, item_index_base, sb_rows, variant, floor,
                           rows_wg, ksel, gstats, ov, oi)
    if caller_order and uop.src is not None:            # (the table-driven tails work in layout order: permute at the end)
        pos = uop.pos.long()
        ov, oi = ov.index_select(0, pos), oi.index_select(0, pos)
    return ov, oi


def _redo_flagged(uop, iop, blockmax, flag, n_bad, n_sb, k, user_bias, item_bias, item_index_base, sb_rows, variant, floor, rows_wg,
                  ksel, gstats, ov, oi, out_index=None):
    """The users the first pass flagged (``n_bad`` of them, ``flag`` != 0): the wide second pass on their table columns, then
    the exact fp32 MFMA path for what is left.  Returns (ov, oi) with their rows replaced.  ``out_index``: operand row r's
    result lives in row out_index[r] of ov / oi (class-sorted users whose lists leave in the caller's order)."""
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad

    def rows_of(b):
        return b if out_index is None else out_index.index_select(0, b).long()
    if n_bad:
        bad = torch.nonzero(flag, as_tuple=False).reshape(-1)
        if N.load().trec_get_tuning(b"topk_filter_wide_pass", 1) != 0 and n_sb > ksel:
            # tier 1: FILTER_KSEL_WIDE slots; tier 2, for the few users still over (and only if they are few: it re-scores ALL
            # their superblocks that reach the floor): every superblock may be kept
            # (trec_topk_collect_blocks takes at most COLLECT_KSEL_MAX slots: on catalogues above 4096 superblocks = 2,097,152
            # items tier 2 keeps that many -- a user wanting more stays flagged and goes to the exact fp32 path)
            for tier, ksel_w in enumerate((min(FILTER_KSEL_WIDE, n_sb), min(n_sb, COLLECT_KSEL_MAX))):
                if bad.numel() == 0 or (tier == 1 and (ksel_w <= FILTER_KSEL_WIDE or bad.numel() > WIDE_TIER2_MAX_FRACTION * n_u)):
                    break
                wv, wi, wflag = _wide_second_pass(uop, iop, blockmax, bad, n_sb, k, user_bias, item_bias, item_index_base,
                                                  sb_rows, variant, floor, rows_wg, max(int(k), ksel_w), gstats)
                ov[rows_of(bad)] = wv
                oi[rows_of(bad)] = wi
                bad = bad[wflag != 0]
                LAST_FILTER_STATS["flagged_after_wide_pass" + ("" if tier == 0 else "_2")] = int(bad.numel())
            LAST_FILTER_STATS["users_on_fp32_fallback"] = int(bad.numel())
    del blockmax
    if n_bad and bad.numel():
        ub = user_bias[bad].contiguous() if user_bias is not None else None
        with _timed("topk_filter_fallback"):
            fv, fi = score_topk(uop.f32[bad].contiguous(), iop.f32, DTYPE_F32, kpad, int(k), ub, item_bias, MODE_DOT,
                                item_index_base=item_index_base, method="two_stage" if n_i >= TWO_STAGE_MIN_ITEMS else "direct")
        ov[rows_of(bad)] = fv
        oi[rows_of(bad)] = fi
    return ov, oi


WIDE_K_MAX = 64                   # largest k of score_topk_filtered_wide
WIDE_CANDIDATES = 1024            # candidate slots per user there (trec_topk_merge takes up to 1024 entries)


def score_topk_filtered_wide(uop, iop, k, user_bias=None, item_bias=None, item_index_base=0, sb_rows=None):
    """EXACT fp32 top-k for 17 <= k <= 64 through the int8 -> bf16 cascade (k > 16 used to take the all-fp32 MFMA
    path, ~17x the cascade's time per pair): stages 0 and 1 of score_topk_filtered as they are -- the int8 pass over every pair,
    tau = the k-th largest of the chunks' lower-bound lists, the compaction, the bf16 refining launch that LISTS every item able to
    reach the top-k (1,024 slots per user here) -- and a finish made of library calls: the reference's fp32 chain on every listed
    pair (trec_pair_score_exact) and the k best of each list (trec_topk_merge).  A list holds every item whose fp32 score reaches
    the k-th best (that argument does not depend on k), so its k best ARE the answer.  Users whose list is incomplete
    (more candidates than slots, no usable bound) are re-done from exact fp32 score slabs and exact ranks; if the int8 bound is
    too loose for the catalogue nobody is listed and everybody is.  ``uop``: score_prep_filter(sort_users=True, k=k).  Single process.
    Returns (values [n_users, k], ids [n_users, k]) in the caller's order, bit-identical to score_topk(..., DTYPE_F32)."""
    kk = int(k)
    if not 16 < kk <= WIDE_K_MAX:
        raise ValueError("score_topk_filtered_wide covers 17 <= k <= %d" % WIDE_K_MAX)
    if uop.kpad not in (64, 128) or iop.gstats is None or uop.wg_rows is None:
        raise ValueError("score_topk_filtered_wide needs class-sorted users (sort_users=True), kpad 64 / 128 and item gstats")
    dev = uop.bf16.device
    sb_rows = int(sb_rows or SUPERBLOCK_ROWS)
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad
    n_sb = (n_i + sb_rows - 1) // sb_rows
    n_real = int(uop.n_real)
    if user_bias is None:
        ub = None
    elif uop.bias_sorted is not None and uop.bias_ref is user_bias:
        ub = uop.bias_sorted
    else:
        ub = user_bias.reshape(-1).index_select(0, uop.perm).masked_fill_(uop.pad, 0.0)
    ib = _f32c(item_bias.detach()).reshape(-1) if item_bias is not None else None
    rows_wg = N.query("trec_score_rows_per_workgroup", DTYPE_BF16, kpad)
    rblocks = (n_u + rows_wg - 1) // rows_wg
    n_chunks = max(-(-kk // 16) + 1, min(n_sb, N.load().trec_get_tuning(b"cascade_max_chunks", CASCADE_MAX_CHUNKS),
                                         -(-32 * 768 // rblocks)))             # k distinct entries need ceil(k / 16) lists
    LAST_FILTER_STATS.clear()
    lib = N.load()
    table = cands = None
    bad = None
    if n_sb >= kk:
        table, _stride, (rows, overflow), _tau, cands = _cascade_stage1(uop, iop, kk, ub, ib, sb_rows, n_sb, n_chunks, None, None,
                                                                         iop.gstats, item_index_base,
                                                                         candidates_cap=WIDE_CANDIDATES)
        del table
    ov = torch.empty((n_u, kk), dtype=torch.float32, device=dev)
    oi = torch.empty((n_u, kk), dtype=torch.int32, device=dev)
    real = uop.src >= 0
    if cands is None:
        bad = real.clone()                                  # (too loose, or fewer superblocks than k: the fp32 path for everybody)
        LAST_FILTER_STATS["prefilter"] = "int8 (too loose: fp32 path)"
    else:
        n = cands.n
        with _timed("topk_wide_finish"):
            # one wave per user: the reference's fp32 chain on every listed item, the k best by value then index; users whose
            # list is incomplete (more entries than slots, fewer than k) get flagged there
            N.call("trec_topk_candidates_finish_wide", N.ptr(n), N.ptr(cands.items), cands.cap, N.ptr(cands.floor0),
                   N.ptr(uop.f32), N.ptr(iop.f32), kpad, kpad, uop.d, N.ptr(ub), N.ptr(ib), item_index_base, n_u, kk,
                   N.ptr(ov), N.ptr(oi), N.ptr(cands.flag), N.ptr(cands.n_flagged), None)
        complete = real & (cands.flag == 0) & torch.isfinite(cands.floor0)
        bad = real & ~complete
        LAST_FILTER_STATS.update({"prefilter": "int8", "refined_rows": int(rows), "tail": "candidate lists, wide finish",
                                  "candidates_cap": cands.cap, "prerefined_pairs": n_real * kk if cands.pre is not None else 0,
                                  "candidates_per_user": float(n.clamp(max=cands.cap)[complete].float().mean().item()) if bool(complete.any()) else 0.0})
    n_bad = int(bad.sum().item())
    if n_bad:
        rows_b = torch.nonzero(bad, as_tuple=False).reshape(-1)
        # (the fused fp32 top-k kernels hold 16 entries per list: these users take exact fp32 score slabs + exact ranks)
        step = max(1, (1 << 28) // max(1, n_i))
        with _timed("topk_filter_fallback"):
            for b0 in range(0, n_bad, step):
                rb = rows_b[b0:b0 + step]
                slab = score_store(uop.f32[rb].contiguous(), iop.f32, DTYPE_F32, kpad, ub[rb].contiguous() if ub is not None else None,
                                   ib, MODE_DOT)
                fv, fi = topk_from_scores(slab, kk)
                ov[rb] = fv
                oi[rb] = torch.where(fi >= 0, fi + int(item_index_base), fi)
                del slab
    pos = uop.pos.long()                                    # the caller's rows
    LAST_FILTER_STATS.update({"users": n_real, "layout_rows": int(n_u), "flagged_users": n_bad, "route": "cascade, k up to %d" % WIDE_K_MAX})
    return ov.index_select(0, pos), oi.index_select(0, pos)


EUCLID_CANDIDATES = 16       # K' of the Euclidean route: the cascade's largest list
EUCLID_WIDE_K_MAX = 48       # ... and through the WIDE cascade's lists (score_topk_filtered_wide): K' = 32 for 13 <= k <= 24, 64 up to 48


def euclid_candidates_for(k):
    """K' -- how many nearest items the Euclidean route lists per user for the first k places (a third more than k)."""
    kk = int(k)
    if kk <= EUCLID_CANDIDATES - 4:
        return EUCLID_CANDIDATES
    if kk <= 24:
        return 32
    if kk <= EUCLID_WIDE_K_MAX:
        return 64
    raise ValueError("the filtered Euclidean top-k supports k <= %d" % EUCLID_WIDE_K_MAX)
EUCLID_SECOND_PASS_MIN = 32  # users without a certificate from which the cascade runs once more with a measured weight (fewer: the fp32 path)
EUCLID_LAMBDA_PCT = 70       # weight of the item bias in its ordering, in percent of the typical distance sqrt(mean r_u + mean r_i): the NEAREST
                             # items sit closer than the typical one, and a weight above their distance moves the bound's maximum to the
                             # smallest bias (measured, 700 x 280,000, d = 128: sigma_b = 0.02 -> 53 / 0 / 155 users re-done at 100 / 70 / 25 %;
                             # sigma_b = 0.2 -> 700 / 201 / 700; profiles/r06_euclid_bias_ab.json)


EUCLID_LAMBDA_SAMPLE_MIN = 8192   # users from which the bias weight is MEASURED on a sample of 512 users first (below: the guess, then the second pass)


def score_topk_euclid_filtered(user_repr, item_repr, k, user_bias=None, item_bias=None, item_index_base=0, _lam=None, _kc=None,
                               _measure=False):
    """EXACT top-k of the Euclidean scores -sqrt(max(r_u - 2 u.i + r_i, 1e-16)) (+ biases) -- prediction_graphs.py:84-100 +
    recommendation_graphs.py:33-41, :73-82 -- through the DOT-product cascade (csrc/euclid_topk.hip): per user, nearest = largest
    g = u.i - r_i / 2, so the cascade runs with the item "bias" -r_i / 2 and lists the K' = 16 nearest items (k <= 12; K' = 32 / 64
    through the wide cascade for 13 <= k <= 48, int8-cascade catalogues only); the reference's own
    chain re-scores those pairs (trec_pair_score_exact: the oracle's bits, biases included); a per-user certificate -- no item
    outside the K' can reach the first k places, given the K'-th largest g and the largest item bias -- decides whether the first
    k of them ARE the answer; users without it (item biases outweighing the distance gap, near-ties) are re-done on the exact
    fp32 MFMA path.  Values and ids are bit-identical to score_topk(..., DTYPE_F32, MODE_EUCLIDEAN) either way.
    Returns (values [U, k], ids [U, k]); LAST_FILTER_STATS["euclid_uncertified_users"] counts the re-done users."""
    kk = int(k)
    if kk < 1:
        raise ValueError("the filtered Euclidean top-k needs k >= 1")
    kc = int(_kc) if _kc is not None else euclid_candidates_for(kk)   # K': 16 (k <= 12) from the cascade's fused lists, 32 / 64 from the wide route's
    u = _f32c(user_repr.detach())
    v = _f32c(item_repr.detach())
    n_u, d = u.shape
    n_i = v.shape[0]
    dev = u.device
    if kc > EUCLID_CANDIDATES and not (cascade_prefilter_for(d, n_i) == "int8" and i8_user_classes_enabled()):
        raise ValueError("the filtered Euclidean top-k with k > %d needs a catalogue the int8 cascade runs on" % (EUCLID_CANDIDATES - 4))
    u32, u_sq, kpad = score_prep(u, DTYPE_F32, want_sqnorm=True)
    i32, i_sq, _ = score_prep(v, DTYPE_F32, want_sqnorm=True)
    c = i_sq * -0.5                                                   # exact halving: the item "bias" of the g ordering
    ub = _f32c(user_bias.detach()).reshape(-1) if user_bias is not None else None
    ib = _f32c(item_bias.detach()).reshape(-1) if item_bias is not None else None
    # item biases in the ordering: h = u.i - r_i / 2 + lambda b_i with lambda = the typical distance of unrelated rows (any lambda
    # >= 0 is valid: the certificate's bound follows it; tuning euclid_bias_in_order = 0 gives the plain g ordering back).  On the
    # device: no host read.
    lam = bmin = None
    if ib is not None and N.load().trec_get_tuning(b"euclid_bias_in_order", 1) != 0:
        lam = _lam if _lam is not None else \
            (torch.sqrt(u_sq.mean() + i_sq.mean()) *
             (N.load().trec_get_tuning(b"euclid_lambda_pct", EUCLID_LAMBDA_PCT) / 100.0)).reshape(1).contiguous()
        lam_source = "given" if _lam is not None else "guess"
        if _lam is None and n_u >= EUCLID_LAMBDA_SAMPLE_MIN and N.load().trec_get_tuning(b"euclid_lambda_sample", 1) != 0:
            # the guess is a fraction of the TYPICAL distance; what counts is how far the NEAREST items are, and the optimum is sharp
            # (profiles/r06_euclid_bias_ab.json): 512 users go through the cascade first and the median distance of their candidates
            # is the weight of the call (~2 ms; before, every eighth user of a 200,000 x 1M call needed the second pass below)
            sel = torch.arange(0, n_u, n_u // 512, device=dev)[:512]
            lam = score_topk_euclid_filtered(u[sel].contiguous(), v, kk, ub[sel].contiguous() if ub is not None else None, ib,
                                             _lam=lam, _kc=kc, _measure=True)
            lam_source = "sampled"
        c = c + lam * ib
        bmin = ib.min().reshape(1)
    prefilter = cascade_prefilter_for(d, n_i)
    i_f = score_prep_filter(v, bias=c, want_gstats=True)
    if kc > EUCLID_CANDIDATES:
        # 13 <= k <= 48: the K' = 32 / 64 largest h through the wide cascade (lists of up to 1,024 candidates, the wave-per-user finish)
        u_f = score_prep_filter(u, sort_users=True, k=kc)
        gv, gi = score_topk_filtered_wide(u_f, i_f, kc, None, c, item_index_base=0)
    else:
        u_f = score_prep_filter(u, sort_users=prefilter == "int8", k=kc)
        gv, gi = score_topk_filtered(u_f, i_f, kc, None, c, item_index_base=0, prefilter=prefilter)
    gv, gi = gv.contiguous(), gi.contiguous()
    stats = dict(LAST_FILTER_STATS)
    # ---- the reference's chain on the U x K' candidate pairs
    xu32 = torch.arange(n_u, dtype=torch.int32, device=dev).repeat_interleave(kc)
    xi32 = gi.reshape(-1).clamp(min=0).contiguous()
    exact = pair_scores_exact(u32, i32, kpad, d, xu32, xi32, ub, ib, MODE_EUCLIDEAN, u_sq, i_sq)

    def measured_weight():
        near = ib[xi32.long()].reshape(n_u, kc) - exact.reshape(n_u, kc)      # = sqrt(D) - b_u
        if ub is not None:
            near = near + ub.reshape(-1, 1)
        return near[gi >= 0].median().clamp(min=0.0).reshape(1).contiguous()

    if _measure:
        return measured_weight()                                              # (the sampling call above: the weight, nothing else)
    bmax = ib.max().reshape(1) if ib is not None else None
    ov = torch.empty((n_u, kk), dtype=torch.float32, device=dev)
    oi = torch.empty((n_u, kk), dtype=torch.int32, device=dev)
    flag = torch.empty((n_u,), dtype=torch.int32, device=dev)
    n_flagged = zero_block(1, dev)
    with _timed("topk_euclid_certify"):
        N.call("trec_topk_euclid_certify", N.ptr(gi), N.ptr(gv), N.ptr(exact), kc, kk, N.ptr(u_sq), N.ptr(ub),
               N.ptr(i_f.gstats), N.ptr(bmax), int(d), n_u, N.ptr(ov), N.ptr(oi), N.ptr(flag), N.ptr(n_flagged), N.ptr(lam),
               N.ptr(bmin))
    n_bad = int(n_flagged.item())
    n_second = 0
    second_pass = False
    wider = (kc < 64 and cascade_prefilter_for(d, n_i) == "int8" and i8_user_classes_enabled() and
             N.load().trec_get_tuning(b"euclid_second_pass_wider", 1) != 0)
    # (a weight measured on a sample and no larger K' to go to: a second pass would repeat the first)
    if n_bad >= EUCLID_SECOND_PASS_MIN and lam is not None and _lam is None and (wider or lam_source != "sampled"):
        # the weight was a guess (a fraction of the TYPICAL distance); the candidates just re-scored say how far the NEAREST items
        # really are: the users without a certificate take the cascade once more with the median of those distances as the weight
        # (its optimum is sharp: profiles/r06_euclid_bias_ab.json) before anybody goes to the fp32 path
        # ... and with the next larger K' where the wide cascade runs (16 -> 32 -> 64): the gap to the K'-th nearest item grows
        second_pass = True
        bad = torch.nonzero(flag, as_tuple=False).reshape(-1)
        lam2 = measured_weight()
        kc2 = 2 * kc if wider else kc
        sv, si = score_topk_euclid_filtered(u[bad].contiguous(), v, kk, ub[bad].contiguous() if ub is not None else None, ib,
                                            item_index_base=0, _lam=lam2, _kc=kc2)
        n_second = n_bad - int(LAST_FILTER_STATS.get("euclid_uncertified_users", 0))
        ov[bad] = sv
        oi[bad] = si
        n_bad = 0                                           # (the second pass finished its own leftovers on the fp32 path)
    if n_bad:
        bad = torch.nonzero(flag, as_tuple=False).reshape(-1)
        with _timed("topk_filter_fallback"):
            if kk <= 16:
                fv, fi = score_topk(u32[bad].contiguous(), i32, DTYPE_F32, kpad, kk, ub[bad].contiguous() if ub is not None else None,
                                    ib, MODE_EUCLIDEAN, u_sq[bad].contiguous(), i_sq,
                                    method="two_stage" if n_i >= TWO_STAGE_MIN_ITEMS else "direct")
                ov[bad] = fv
                oi[bad] = fi
            else:
                # (the fused fp32 top-k kernels hold 16 entries per list: exact fp32 score slabs and the k best of every row)
                step = max(1, (1 << 28) // max(1, n_i))
                for b0 in range(0, n_bad, step):
                    rb = bad[b0:b0 + step]
                    slab = score_store(u32[rb].contiguous(), i32, DTYPE_F32, kpad, ub[rb].contiguous() if ub is not None else None,
                                       ib, MODE_EUCLIDEAN, u_sq[rb].contiguous(), i_sq)
                    fv, fi = topk_from_scores(slab, kk)
                    ov[rb] = fv
                    oi[rb] = fi
                    del slab
    if item_index_base:
        oi = torch.where(oi >= 0, oi + int(item_index_base), oi)
    LAST_FILTER_STATS.clear()
    LAST_FILTER_STATS.update(stats)
    if second_pass:
        n_bad = int(n_flagged.item()) - n_second            # users that reached the fp32 path in the end
    LAST_FILTER_STATS.update({"route": "euclidean via the dot-product cascade (g = u.i - r_i / 2, %d nearest, certificate)" % kc,
                              "users": n_u, "euclid_uncertified_users": n_bad, "euclid_second_pass_certified": n_second,
                              "euclid_bias_weight": lam_source if lam is not None else "none"})
    return ov, oi


def topk_merge(cand_vals, cand_idx, k):
    n_u, n_cand = cand_vals.shape
    ov = torch.empty((n_u, k), dtype=torch.float32, device=cand_vals.device)
    oi = torch.empty((n_u, k), dtype=torch.int32, device=cand_vals.device)
    N.call("trec_topk_merge", N.ptr(cand_vals), N.ptr(cand_idx), n_u, n_cand, k, N.ptr(ov), N.ptr(oi))
    return ov, oi
