"""
Sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

The reference has no multi-device code at all; this is the MI355X-native scaling path named by
BASELINE.json's north_star.

PREDICT -- items (rows of the item feature matrix, hence of the item representation) are split row-wise, the user tile is
replicated, and small per-user exchanges finish a query:

  * top-k  : every rank runs the exact top-k pipeline on its item shard with global item ids (``item_index_base``).  Large
             catalogues (the filtered / cascade routes) first agree on ONE top-k floor per user -- the k largest lower bounds of
             every shard, exchanged and reduced (``shared_topk_floor*``) -- so that a shard re-scores only what can reach the
             GLOBAL top-k; then the per-shard [U, k] lists are merged.  Both exchanges come in two forms: user-partitioned
             ALL-TO-ALL (``*_a2a``, the default over RCCL: every rank receives 1/world of what an all-gather would deliver and
             finalises ITS users; ``replicate=True`` adds an all-gather of the finished lists where every rank needs them) and
             plain ALL-GATHER (backends without a device all-to-all: the gloo tests).  The merge order (value desc, index asc)
             is a total order over disjoint item ids, so every form ends in the same, exact, global top-k.
  * ranks  : rank = 1 + count of items that beat the target (recommendation_graphs.py:73-82 is a count);
             counts over disjoint item ranges add, so an all-reduce(SUM) of int32 partial counts gives exact ranks.

FIT -- data-parallel over USERS (the reference's own batching axis, tensorrec.py:199-217): each rank owns a slice of user rows
(interactions + user features); the objective is a sum over interactions (a scalar loss all-reduces its sums inside the loss op,
ops.scalar_loss_group), so one synchronous step equals ONE single-process step on the union batch (not the reference's sequential
per-batch steps).  What crosses the wire is decided per weight table by ``plan_gradient_exchange``:

  * "disjoint"   -- the table's gradient rows are touched by one rank only (identity / one-hot user features under user shards):
                    NO exchange, the owner steps its rows (weights and Adam slots of other ranks' rows go stale on this replica
                    until ``TensorRec.dp_sync``);
  * "sharded"    -- large shared tables: reduce-scatter of the gradient by row range -> Adam on the owned rows -> all-gather of
                    the updated rows (each Adam slot lives on ONE rank: 1/world of the optimiser traffic and 7/8 of an
                    all-reduce's bytes at world 8);
  * "replicated" -- small tensors: all-reduce(SUM) + the identical fused Adam on every rank.

The device sampler is keyed by global user id, so shards draw what the whole population would.

No collective sits inside a score kernel; payloads are KBs-MBs against ~10 ms of MFMA work per 65k-user tile, so xGMI (7
point-to-point links x ~153 GB/s) is nowhere near a bound for predict; the fit's exchange (0.9 GB per rank and step at 1M x 1M,
d = 128) is the part that is.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


# Collectives are skipped in a one-rank world -- unless FORCE_COLLECTIVES is set: the one-GPU RCCL smoke test
# (tests/test_gpu_rccl_world1.py) runs every exchange of the scoring / training path through the real "nccl" backend at
# world size 1, which is as much of RCCL as a one-GPU box can execute.
FORCE_COLLECTIVES = False


def active(group=None):
    """True when the multi-rank code paths (and their collectives) run for ``group``."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def shard_bounds(n_items, world_size, rank, align=64):
    """Contiguous, aligned, nearly equal item ranges: [begin, end) of ``rank``.  ``align`` keeps shard starts on the
    score kernel's tile so float4 bias loads stay aligned; the last shard takes the remainder."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    per = -(-n_items // world_size)
    per = -(-per // align) * align
    begin = min(rank * per, n_items)
    end = min(begin + per, n_items)
    return begin, end


def all_gather_cat(t, group=None, dim=1):
    """All-gather equal-shaped tensors and concatenate along ``dim`` (rank order): ONE collective into one buffer."""
    world = dist.get_world_size(group)
    if world == 1 and not FORCE_COLLECTIVES:
        return t
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)           # rank-major along dim 0
    if dim == 0:
        return out
    return torch.cat(list(out.reshape((world,) + tuple(t.shape)).unbind(0)), dim=dim)


def exchange_topk(local_vals, local_idx, group=None):
    """Per-shard [U, k] lists -> candidate tables [U, world*k] on every rank (one all-gather each for values/ids)."""
    return all_gather_cat(local_vals, group), all_gather_cat(local_idx, group)


def merge_topk(cand_vals, cand_idx, k):
    """k best of the gathered candidates, (value desc, index asc).  GPU tensors only: this is the HIP merge kernel."""
    from . import ops
    return ops.topk_merge(cand_vals.contiguous(), cand_idx.contiguous(), k)


def sharded_top_k(local_vals, local_idx, k, group=None):
    """local lists (global item ids) -> exact global top-k, identical on every rank."""
    cv, ci = exchange_topk(local_vals, local_idx, group)
    return merge_topk(cv, ci, k)


def reduce_rank_counts(local_counts, group=None):
    """Partial 'items that beat the target' counts (int32) -> global ranks - 1 on every rank (all-reduce SUM)."""
    if active(group):
        dist.all_reduce(local_counts, op=dist.ReduceOp.SUM, group=group)
    return local_counts


def all_reduce_sum_(tensors, group=None):
    """In-place SUM all-reduce of a list of tensors (weight gradients of the user-sharded fit)."""
    if active(group):
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


def kth_largest_block_max(block_maxima, k):
    """[n_blocks, n_users] superblock maxima -> [n_users] the k-th largest per user (-inf with fewer than k): the
    stage-2 selection kernel run for its floor output only."""
    from . import ops, _native as N
    n_blocks, n_users = block_maxima.shape
    sel = torch.empty((n_users, k), dtype=torch.int32, device=block_maxima.device)
    floor = torch.empty((n_users,), dtype=torch.float32, device=block_maxima.device)
    N.call("trec_topk_select_blocks", N.ptr(block_maxima), n_blocks, n_users, n_users, k, N.ptr(sel), None, N.ptr(floor))
    return floor


def shared_topk_floor(sel_max, group=None):
    """The ``floor_exchange`` of ops.score_topk_two_stage for item shards: every rank contributes the maxima of its k
    selected superblocks ([k, n_users], 4k bytes per user -- 40 MB at 1M users, k = 10); the k-th largest of the
    world * k gathered maxima bounds every user's global k-th best score from below (k disjoint superblocks each hold an
    item scoring at least that).  ONE all-gather; the result is identical on every rank."""
    k = sel_max.shape[0]
    gathered = all_gather_cat(sel_max, group, dim=0) if active(group) else sel_max             # [world * k, n_users]
    return kth_largest_block_max(gathered, k)


# ---- exchanges partitioned by USER ------------------------------------------------------------------------------
# The all-gather forms above leave every rank with every user's result: each rank RECEIVES (world - 1) x the payload
# (8 ranks, 1M users, k = 10: 7 x 40 MB of superblock maxima + 7 x 80 MB of lists per rank).  A rank only has to
# FINALISE its share of the users, so the same information can travel as all-to-all slices -- per rank (world - 1) / world
# of ONE payload (35 MB + 70 MB) -- followed, where every rank needs the result, by an all-gather of the final values
# only (4 bytes per user for the floor, k * 8 bytes per user for the lists).  xGMI is point-to-point, so received bytes
# per rank are what the exchange costs.
def user_slice(n_users, world_size, rank):
    """Contiguous, equal-sized (padded) user ranges: rank r finalises users [r * per, min((r + 1) * per, n_users))."""
    per = -(-n_users // world_size)
    return min(rank * per, n_users), min((rank + 1) * per, n_users), per


def _all_to_all_user_slices(t, user_dim, group):
    """t has all users along ``user_dim``; returns [world, ...] where entry s is rank s's data for THIS rank's users
    (user axis cut to the padded slice length)."""
    world = dist.get_world_size(group)
    n_users = t.shape[user_dim]
    per = -(-n_users // world)
    t = t.movedim(user_dim, 0)
    if per * world != n_users:
        pad = torch.zeros((per * world - n_users,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    send = t.contiguous()                                         # [world * per, ...]: destination-major
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    return recv.reshape((world, per) + tuple(send.shape[1:]))


def exchange_floor_table_a2a(sel_max, group=None):
    """[k, n_users] maxima of this rank's k selected superblocks -> ([world * k, per] table of ALL ranks' maxima for this
    rank's users, (begin, end) of those users)."""
    k, n_users = sel_max.shape
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    got = _all_to_all_user_slices(sel_max, 1, group)              # [world, per, k]
    per = got.shape[1]
    b, e, _ = user_slice(n_users, world, rank)
    return got.permute(0, 2, 1).reshape(world * k, per).contiguous(), (b, e)


def shared_topk_floor_a2a(sel_max, group=None, kth_fn=None):
    """shared_topk_floor with user-partitioned traffic: every rank receives the other ranks' k maxima for ITS users only
    (all-to-all), takes the k-th largest of the world * k values (``kth_fn``, default the stage-2 HIP kernel), and the
    [n_users] floor is completed by an all-gather of the slices (4 bytes per user)."""
    kth_fn = kth_fn or kth_largest_block_max
    k, n_users = sel_max.shape
    if not active(group):
        return kth_fn(sel_max, k)
    world = dist.get_world_size(group)
    table, (b, e) = exchange_floor_table_a2a(sel_max, group)
    per = table.shape[1]
    mine = kth_fn(table, k) if per else table.new_empty((0,))
    mine[e - b:] = float('-inf')                                  # padded users
    out = torch.empty((world * per,), dtype=sel_max.dtype, device=sel_max.device)
    dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
    return out[:n_users].contiguous()


def exchange_topk_a2a(local_vals, local_idx, group=None):
    """Per-shard [n_users, k] lists -> candidate tables [per, world * k] for THIS rank's users (one all-to-all each for
    values / ids) and the (begin, end) of those users."""
    n_users, k = local_vals.shape
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gv = _all_to_all_user_slices(local_vals, 0, group)            # [world, per, k]
    gi = _all_to_all_user_slices(local_idx, 0, group)
    per = gv.shape[1]
    b, e, _ = user_slice(n_users, world, rank)
    if e - b < per:                                               # padded users: empty candidates
        gi[:, e - b:] = -1
        gv[:, e - b:] = float('-inf')
    return (gv.permute(1, 0, 2).reshape(per, world * k).contiguous(),
            gi.permute(1, 0, 2).reshape(per, world * k).contiguous(), (b, e))


def sharded_top_k_a2a(local_vals, local_idx, k, group=None, replicate=False, merge_fn=None):
    """Per-shard [n_users, k] lists (global item ids) -> the exact global top-k of THIS rank's users
    (user_slice(n_users, world, rank)): one all-to-all of list slices + a local merge (``merge_fn``, default the HIP merge
    kernel).  ``replicate``: finish with an all-gather of the merged lists so that every rank holds all users (what
    sharded_top_k returns)."""
    if not active(group):
        return local_vals, local_idx
    merge_fn = merge_fn or merge_topk
    n_users = local_vals.shape[0]
    world = dist.get_world_size(group)
    cv, ci, (b, e) = exchange_topk_a2a(local_vals, local_idx, group)
    per = cv.shape[0]
    mv, mi = merge_fn(cv, ci, k)
    if not replicate:
        return mv[:e - b], mi[:e - b]
    ov = torch.empty((world * per, k), dtype=mv.dtype, device=mv.device)
    oi = torch.empty((world * per, k), dtype=mi.dtype, device=mi.device)
    dist.all_gather_into_tensor(ov, mv.contiguous(), group=group)
    dist.all_gather_into_tensor(oi, mi.contiguous(), group=group)
    return ov[:n_users], oi[:n_users]


_A2A_CHECKED = {}          # process group -> the all-to-all known-answer check passed (agreed over the ranks)


def a2a_available(t, group=None):
    """All-to-all of device tensors needs RCCL ("nccl"); gloo carries CUDA tensors only for the gather / reduce
    collectives (the one-GPU functional tests).  CPU tensors: gloo has all_to_all_single.
    The first call per process group runs a known-answer all-to-all (a COLLECTIVE: every rank reaches this point, as every
    rank reaches the exchange it guards) and the ranks agree on the outcome; a wrong answer -- or TREC_SHARD_EXCHANGE=allgather
    in the environment -- keeps the all-gather forms for the life of the process (the public API used to trust the
    backend name alone)."""
    import os
    if not active(group):
        _A2A_CHECKED.clear()                                      # (no process group, or a one-rank world: nothing to remember)
        return False
    if t.is_cuda and dist.get_backend(group) != "nccl":
        return False
    if os.environ.get("TREC_SHARD_EXCHANGE", "").lower() == "allgather":
        return False
    # (keyed by the group AND what it is made of: after destroy_process_group + a new init a recycled id, another backend or
    # another world size must not meet a stale verdict -- ADVICE r3)
    key = (id(group) if group is not None else 0, dist.get_backend(group), dist.get_world_size(group), dist.get_rank(group),
           str(t.device))
    for stale in [k_ for k_ in _A2A_CHECKED if k_[0] == key[0] and k_ != key]:
        del _A2A_CHECKED[stale]
    if key not in _A2A_CHECKED:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        x = torch.arange(world * 3, dtype=torch.float32, device=t.device) + 100.0 * rank
        want = torch.stack([torch.arange(rank * 3, rank * 3 + 3, dtype=torch.float32) + 100.0 * s_ for s_ in range(world)])
        try:
            good = torch.equal(_all_to_all_user_slices(x, 0, group).cpu(), want)
        except RuntimeError:                                      # (a backend without all_to_all raises on every rank alike)
            good = False
        ok = torch.tensor([1.0 if good else 0.0], device=t.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)    # every rank must take the same exchange form
        _A2A_CHECKED[key] = bool(ok.item() == 1.0)
    return _A2A_CHECKED[key]


def collective_selfcheck(device, group=None):
    """Tiny known-answer run of every collective the scoring / training path uses (all-gather, all-to-all, all-reduce SUM
    / MAX): the first thing bench.py does on a multi-GPU launch, so a broken RCCL / xGMI setup fails loudly before any
    number is reported.  Returns 'ok', or -- identically on every rank -- a message when only the all-to-all answer is
    wrong (the caller then keeps the all-gather forms of the exchanges)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    x = torch.arange(world * 3, dtype=torch.float32, device=device) + 100.0 * rank
    g = all_gather_cat(x.reshape(1, -1), group, dim=0)
    want = torch.stack([torch.arange(world * 3, dtype=torch.float32) + 100.0 * r for r in range(world)]).to(device)
    assert torch.equal(g, want), "all-gather self-check failed"
    import os
    a2a_expected = (x.is_cuda and dist.get_backend(group) == "nccl" or not x.is_cuda) and \
        os.environ.get("TREC_SHARD_EXCHANGE", "").lower() != "allgather"          # (the user's own choice is not a failure)
    if a2a_expected and not a2a_available(x, group):
        return "all-to-all self-check failed: falling back to the all-gather exchange"
    s = torch.tensor([rank + 1.0, 10.0], device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    assert s.tolist() == [world * (world + 1) / 2.0, 10.0 * world], "all-reduce(SUM) self-check failed"
    m = all_reduce_max(torch.tensor([float(rank)], device=device), group)
    assert m.item() == world - 1.0, "all-reduce(MAX) self-check failed"
    return "ok"


def all_reduce_max(t, group=None):
    """MAX all-reduce of a small float tensor (the item-side maxima behind the bf16 filter's error bound: the bound must
    cover the items of EVERY shard, ops.score_topk_filtered)."""
    if active(group):
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_reduce_scalar(value, device, group=None):
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    if active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def max_over_ranks(seconds, device, group=None):
    """Timing helper for bench.py: the slowest rank defines the step time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if active(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


# ---- gradient exchange of the data-parallel fit -------------------------------------------------------------------------
# Round 3 all-reduced EVERY weight gradient densely each step (1.02 GB of payload at 1M x 1M, d = 128: ~1.8 GB on the wire per
# rank) and ran the identical dense Adam on every rank.  A step now moves only what a rank cannot know by itself:
#
#   "disjoint"    the rows of the table a rank's gradient can touch -- the column support of the feature matrices the variable
#                 is multiplied with (ops._note_row_support; identity / one-hot USER features under user sharding) -- do not
#                 overlap between ranks and follow the rank order: rank r owns [b_r, b_{r+1}) (its support plus the untouched
#                 rows up to the next rank's), holds the complete gradient of those rows already, and steps Adam on them
#                 alone.  NOTHING is exchanged per step; the other ranks' rows of this replica go stale until
#                 ``sync_owned_rows`` (end of the fit call) broadcasts every owner's rows, weights and Adam slots.
#   "sharded"     any other large table (the ITEM side: every rank's users touch every item row): reduce-scatter(SUM) of the
#                 gradient -> Adam on the owned 1/world of the rows -> all-gather of the updated rows.  The same bytes on the wire
#                 as an all-reduce, but the optimiser runs once per row instead of world times, its slots for the other rows
#                 are never touched, and the exchange of the item table runs on RCCL's stream while the user side's Adam runs.
#   "replicated"  small tensors (bias vectors, the dense layers of a ReLU tower): all-reduce + the identical Adam everywhere.
#
# Floor for a synchronous, step-equivalent update of a replicated table of S bytes: every rank must receive the other ranks'
# contributions to the rows it steps ((world - 1) / world x S) and every updated row it does not own (the same again) --
# 0.9 GB per rank and step for the 512 MB item table at world 8; only staleness or narrower wire formats go below, and neither
# keeps the step equal to the single-process one.
SHARD_MIN_NUMEL = 1 << 18      # tensors below this many elements are all-reduced and stepped everywhere


class GradPlan(object):
    """How each variable's gradient travels and which rows this rank steps: ``mode[name]`` in {"replicated", "sharded",
    "disjoint"}, ``own[name]`` = (lo, hi) rows this rank owns, ``bounds[name]`` = the world + 1 row boundaries of all ranks."""

    def __init__(self):
        self.mode, self.own, self.bounds, self.key = {}, {}, {}, None

    def wire_bytes_per_step(self, shapes):
        """(sent = received) bytes per rank and step, by variable: the ring / direct forms of both collectives move
        (world - 1) / world of the payload per rank each way."""
        out = {}
        for name, mode in self.mode.items():
            numel = 1
            for s in shapes[name]:
                numel *= int(s)
            world = len(self.bounds[name]) - 1 if name in self.bounds else 1
            f = (world - 1) / float(max(world, 1))
            out[name] = {"disjoint": 0.0, "sharded": 2.0 * f * numel * 4, "replicated": 2.0 * f * numel * 4}[mode]
        return out


def plan_gradient_exchange(names, shapes, supports, device, group=None):
    """A COLLECTIVE (one small all-gather): every rank passes, per variable, the row support of its gradient ((lo, hi) or
    None = any row) and all ranks derive the same plan.  ``shapes[name]``: the variable's shape (rows = dim 0)."""
    plan = GradPlan()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if active(group) else (1, 0)
    mine = torch.full((len(names), 2), -1, dtype=torch.int64, device=device)
    for j, name in enumerate(names):
        sup = supports.get(name)
        if sup is not None:
            mine[j, 0], mine[j, 1] = int(sup[0]), int(sup[1])
    allsup = all_gather_cat(mine.reshape(1, -1), group, dim=0).reshape(world, len(names), 2).cpu().numpy() if active(group) \
        else mine.reshape(1, len(names), 2).cpu().numpy()
    for j, name in enumerate(names):
        rows = int(shapes[name][0]) if len(shapes[name]) else 1
        numel = 1
        for s in shapes[name]:
            numel *= int(s)
        sup = allsup[:, j, :]
        mode = "replicated"
        bounds = None
        if world > 1 or FORCE_COLLECTIVES:
            if (sup >= 0).all():
                lo, hi = sup[:, 0].copy(), sup[:, 1].copy()
                prev = 0
                ok = True
                for r in range(world):
                    if hi[r] <= lo[r]:                      # an empty support: owns nothing of its own
                        lo[r] = hi[r] = prev
                    if lo[r] < prev or hi[r] > rows:
                        ok = False
                        break
                    prev = hi[r]
                if ok:
                    mode = "disjoint"
                    bounds = [0] + [int(lo[r]) for r in range(1, world)] + [rows]
            if mode == "replicated" and numel >= SHARD_MIN_NUMEL and rows >= world:
                mode = "sharded"
                per = -(-rows // world)
                bounds = [min(r * per, rows) for r in range(world)] + [rows]
        plan.mode[name] = mode
        if bounds is not None:
            plan.bounds[name] = bounds
            plan.own[name] = (bounds[rank], bounds[rank + 1])
    plan.key = tuple((n, plan.mode[n], tuple(plan.bounds.get(n, ()))) for n in names)
    return plan


def _rs_supported(t, group):
    return dist.get_backend(group) == "nccl"           # (gloo has no reduce-scatter at all)


def reduce_scatter_rows(grad, bounds, rank, group=None, async_op=False):
    """SUM over the ranks of ``grad`` [rows, ...], this rank receiving rows [bounds[rank], bounds[rank + 1]) (equal-sized
    ranges of ``per`` rows; the last may be short).  Returns (own_rows_tensor, work or None).  RCCL: reduce_scatter_tensor (the
    gradient is padded only when the rows do not divide); gloo (functional tests on one GPU / CPU): an all-reduce and a slice."""
    world = len(bounds) - 1
    rows = grad.shape[0]
    lo, hi = bounds[rank], bounds[rank + 1]
    if not _rs_supported(grad, group):
        work = dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return grad[lo:hi], work
    per = -(-rows // world)
    src = grad
    if per * world != rows:
        src = torch.zeros((per * world,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
        src[:rows].copy_(grad)
    out = torch.empty((per,) + tuple(grad.shape[1:]), dtype=grad.dtype, device=grad.device)
    work = dist.reduce_scatter_tensor(out, src.contiguous(), op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return out[:hi - lo], work


def all_gather_rows(weights, bounds, rank, group=None, async_op=False):
    """Every rank's owned rows of ``weights`` [rows, ...] (equal-sized ranges) into every replica, in place."""
    world = len(bounds) - 1
    rows = weights.shape[0]
    lo, hi = bounds[rank], bounds[rank + 1]
    per = -(-rows // world)
    if per * world == rows and dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(weights, weights[lo:hi], group=group, async_op=async_op)      # in place
    mine = torch.zeros((per,) + tuple(weights.shape[1:]), dtype=weights.dtype, device=weights.device)
    mine[:hi - lo].copy_(weights[lo:hi])
    full = torch.empty((per * world,) + tuple(weights.shape[1:]), dtype=weights.dtype, device=weights.device)
    dist.all_gather_into_tensor(full, mine, group=group)
    with torch.no_grad():
        weights.copy_(full[:rows])
    return None


def sync_owned_rows(tensors, bounds, group=None):
    """Every owner's rows [bounds[r], bounds[r + 1]) of each tensor in ``tensors`` to every replica (one broadcast per owner
    and tensor: the ranges of a "disjoint" plan differ in size).  Called at the end of a fit call."""
    world = len(bounds) - 1
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        if hi <= lo:
            continue
        src = dist.get_global_rank(group, r) if group is not None else r
        for t in tensors:
            with torch.no_grad():
                seg = t[lo:hi]
                if seg.is_contiguous():
                    dist.broadcast(seg, src=src, group=group)
                else:                                   # pragma: no cover  (rows of a contiguous tensor are contiguous)
                    tmp = seg.contiguous()
                    dist.broadcast(tmp, src=src, group=group)
                    seg.copy_(tmp)
