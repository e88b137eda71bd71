"""What ONE rank of an 8-GPU item-sharded predict does in the EXACT mode, timed on one GPU: user-side K1 + operand prep for
all 1M users (replicated), its 125k-item shard, the filtered top-10 with the shared floor (emulated: the 2nd largest local
superblock maximum stands in for the k-th largest over all shards, so ~1/8 of the kept superblocks stay local), then the
merge of the 8 x 10 candidates of the U/8 users this rank finalises (what the all-to-all delivers)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops, sharding
from tensorrec_amd.sparse import SparseFeatures

W = 8
U, I, d, k = 1_000_000, 1_000_000 // W, 128, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
w_u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
w_i = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
f_u = SparseFeatures(sp.identity(U, dtype=np.float32, format="csr"), "cuda")
f_i = SparseFeatures(sp.identity(I, dtype=np.float32, format="csr"), "cuda")
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")

def floor_exchange(sel_max):            # [k, U] sorted desc per user: row 1 = 2nd largest
    return sel_max[1].contiguous()

def step():
    u = ops.spmm_raw(f_u.indptr, f_u.indices, f_u.values, None, U, f_u.nnz, w_u)
    v = ops.spmm_raw(f_i.indptr, f_i.indices, f_i.values, None, I, f_i.nnz, w_i)
    u_f = ops.score_prep_filter(u, sort_users=True, k=k, user_bias=ub)   # users sorted by int8 scale class, as predict_top_k does
    i_f = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    vals, idx = ops.score_topk_filtered(u_f, i_f, k, ub, ib, item_index_base=0, floor_exchange=floor_exchange,
                                        stats_exchange=lambda s: s, prefilter=os.environ.get("PREFILTER", "int8") or None)
    per = U // W                          # the all-to-all leaves this rank with W lists for each of ITS U / W users
    cand_v = vals[:per].repeat(1, W); cand_i = idx[:per].repeat(1, W)
    return sharding.merge_topk(cand_v, cand_i + torch.arange(W, device="cuda").repeat_interleave(k)[None, :] * I, k)

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
