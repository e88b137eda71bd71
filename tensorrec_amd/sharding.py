"""
Item sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

The reference has no multi-device code at all; this is the MI355X-native scaling path named by
BASELINE.json's north_star: items (rows of the item feature matrix, hence of the item representation) are split
row-wise, the user tile is replicated, and exactly ONE small collective finishes a query:

  * top-k  : every rank runs the fused score+top-k kernel on its item shard with global item ids
             (``item_index_base``), then an all-gather of the per-rank [U, k] lists (U*k*8 bytes per rank -- 5 MB
             for 65,536 users, k = 10) and a local k-way merge (trec_topk_merge).  Every rank ends with the same,
             exact, global top-k: the merge order (value desc, index asc) is a total order over disjoint item ids.
  * ranks  : rank = 1 + count of items that beat the target (recommendation_graphs.py:73-82 is a count);
             counts over disjoint item ranges add, so an all-reduce(SUM) of int32 partial counts gives exact ranks.

Training is data-parallel over USERS (the reference's own batching axis, tensorrec.py:199-217): each rank owns a slice of
user rows (interactions + user features), items and all weights are replicated, and one step = local forward/backward +
all-reduce(SUM) of the weight gradients + the identical fused Adam on every rank.  Because the WMRB objective is a sum
over interactions, this equals ONE single-process step on the union batch (not the reference's sequential per-batch
steps).  The device sampler is keyed by global user id, so shards draw what the whole population would.

No collective sits inside the score kernel; payloads are KBs-MBs against ~10 ms of MFMA work per 65k-user tile, so
xGMI (7 point-to-point links x ~153 GB/s) is nowhere near a bound and a plain all-gather is the right primitive.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


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
    if world == 1:
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
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_counts, op=dist.ReduceOp.SUM, group=group)
    return local_counts


def all_reduce_sum_(tensors, group=None):
    """In-place SUM all-reduce of a list of tensors (weight gradients of the user-sharded fit)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
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
    gathered = all_gather_cat(sel_max, group, dim=0) if dist.is_initialized() else sel_max     # [world * k, n_users]
    return kth_largest_block_max(gathered, k)


def all_reduce_max(t, group=None):
    """MAX all-reduce of a small float tensor (the item-side maxima behind the bf16 filter's error bound: the bound must
    cover the items of EVERY shard, ops.score_topk_filtered)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_reduce_scalar(value, device, group=None):
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def max_over_ranks(seconds, device, group=None):
    """Timing helper for bench.py: the slowest rank defines the step time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
