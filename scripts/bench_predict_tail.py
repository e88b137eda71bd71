"""The exact top-k pipeline behind stage 1 at 1M x 1M (or N x N), per-kernel HIP-event times: for A/B of the tail kernels."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops
U = I = int(os.environ.get("N", 1_000_000)); d, k = 128, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
def step():
    u_f = ops.score_prep_filter(u)
    i_f = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    return ops.score_topk_filtered(u_f, i_f, k, ub, ib)
step()
ops.KERNEL_EVENTS = []
for _ in range(3): out = step()
torch.cuda.synchronize()
ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
dur = {}
for n, s, e in ev: dur.setdefault(n, []).append(s.elapsed_time(e))
res = {n: round(float(np.mean(x)), 3) for n, x in dur.items()}
res["tail_total"] = round(sum(x for n, x in res.items() if n != "score_gemm_blockmax"), 3)
res["filter"] = dict(ops.LAST_FILTER_STATS)
print(json.dumps(res))
