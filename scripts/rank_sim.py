"""What ONE rank of an 8-GPU item-sharded predict does, timed on one GPU: 1M users x 125k items (its shard), two-stage
top-10 with a shared floor that prunes ~4/5 of the selected superblocks (emulated: the 2nd largest local superblock
maximum stands in for the all-gathered k-th largest), then the merge of 8 x 10 gathered candidates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorrec_amd import ops, sharding

U, I, d, k = 1_000_000, 125_000, 128, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
u_op, _, kpad = ops.score_prep(u, ops.DTYPE_BF16); v_op, _, _ = ops.score_prep(v, ops.DTYPE_BF16)

def floor_exchange(sel_max):            # [k, U] sorted desc per user: row 1 = 2nd largest
    return sel_max[1].contiguous()

def step():
    vals, idx = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_BF16, kpad, k, ub, ib, item_index_base=0,
                                         floor_exchange=floor_exchange)
    cand_v = vals.repeat(1, 8); cand_i = idx.repeat(1, 8)          # stands in for the all-gathered lists (same size)
    return sharding.merge_topk(cand_v, cand_i + torch.arange(8, device="cuda").repeat_interleave(k)[None, :] * I, k)

for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("per-rank step at N=8 (emulated): %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
