"""What ONE rank of an 8-GPU item-sharded predict does in the EXACT mode, timed on one GPU: user-side K1 + operand prep for
all 1M users (replicated), its 125k-item shard, the filtered top-10 with the SHARED floor, then the merge of the 8 x 10
candidates of the U/8 users this rank finalises (what the all-to-all delivers).
The shared floor is the real one: an untimed first phase runs the int8 stage of all 8 shards (one after the other, same users,
each shard its own items), keeps every shard's k largest lower bounds per user and takes the k-th largest of the 8 k -- exactly
what sharding.shared_topk_floor_a2a leaves on every rank; the timed phase replays shard 0 with that floor."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops, sharding
from tensorrec_amd.sparse import SparseFeatures

W = 8
U, I, d, k = 1_000_000, 1_000_000 // W, 128, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
w_u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
w_i_all = ops.l2_normalize_rows(torch.randn((W * I, d), device="cuda", generator=g))
w_i = w_i_all[:I].contiguous()
f_u = SparseFeatures(sp.identity(U, dtype=np.float32, format="csr"), "cuda")
f_i = SparseFeatures(sp.identity(I, dtype=np.float32, format="csr"), "cuda")
ub = 0.05 * torch.randn(U, device="cuda", generator=g); ib_all = 0.05 * torch.randn(W * I, device="cuda", generator=g)
ib = ib_all[:I].contiguous()
RECORDED, GLOBAL_FLOOR = [], [None]

def floor_exchange(sel_max):            # [k, U]: this shard's k largest lower bounds per user (layout order: the same on every shard)
    if GLOBAL_FLOOR[0] is None:
        RECORDED.append(sel_max.clone())
        return sharding.kth_largest_block_max(sel_max.contiguous(), k)     # (phase 1: the local floor, result unused)
    return GLOBAL_FLOOR[0]

def step():
    u = ops.spmm_raw(f_u.indptr, f_u.indices, f_u.values, None, U, f_u.nnz, w_u)
    v = ops.spmm_raw(f_i.indptr, f_i.indices, f_i.values, None, I, f_i.nnz, w_i)
    u_f = ops.score_prep_filter(u, sort_users=True, k=k, user_bias=ub)   # users sorted by int8 scale class, as predict_top_k does
    i_f = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    vals, idx = ops.score_topk_filtered(u_f, i_f, k, ub, ib, item_index_base=0, floor_exchange=floor_exchange,
                                        stats_exchange=lambda s: s, prefilter=os.environ.get("PREFILTER", "int8") or None,
                                        finish_lanes=int(os.environ.get("FINISH_LANES", "16")))
    per = U // W                          # the all-to-all leaves this rank with W lists for each of ITS U / W users
    cand_v = vals[:per].repeat(1, W); cand_i = idx[:per].repeat(1, W)
    return sharding.merge_topk(cand_v, cand_i + torch.arange(W, device="cuda").repeat_interleave(k)[None, :] * I, k)

for shard in range(W):                   # phase 1 (untimed): every shard's lower bounds -> the floor all ranks would share
    w_i = w_i_all[shard * I:(shard + 1) * I].contiguous(); ib = ib_all[shard * I:(shard + 1) * I].contiguous()
    step()
GLOBAL_FLOOR[0] = sharding.kth_largest_block_max(torch.cat(RECORDED, dim=0).contiguous(), k)
RECORDED.clear()
w_i = w_i_all[:I].contiguous(); ib = ib_all[:I].contiguous()
for _ in range(2): step()
ops.KERNEL_EVENTS = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
dur = {}
for n, s, e in ev: dur.setdefault(n, []).append(s.elapsed_time(e))
out = {"per_rank_step_ms_at_N8_emulated": dt, "kernels_ms": {n: float(np.mean(v)) for n, v in dur.items()},
       "filter": dict(ops.LAST_FILTER_STATS),
       "exchange_bytes_received_per_rank": {"floor all-to-all": (W - 1) * k * (U // W) * 4 + U * 4 * (W - 1) // W,
                                            "lists all-to-all": (W - 1) * (U // W) * k * 8}}
print(json.dumps(out))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "rank_sim.json"), "w"), indent=1)
