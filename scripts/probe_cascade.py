"""The int8 -> bf16 -> fp32 cascade beside the bf16 filter on one box at the bench shape: same operands, exact equality of
the two results, per-kernel times."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops

U = int(os.environ.get("U", 1_000_000)); I = int(os.environ.get("I", 1_000_000)); d = int(os.environ.get("D", 128)); k = 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
out = {}
res = {}
for name, pre in (("bf16_filter", None), ("int8_cascade", "int8")):
    def step():
        uop = ops.score_prep_filter(u); iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
        return ops.score_topk_filtered(uop, iop, k, ub, ib, prefilter=pre)
    step(); torch.cuda.synchronize()
    ops.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    n = 3
    for _ in range(n): r = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dur = {}
    for nm, s, e in ev: dur.setdefault(nm, []).append(s.elapsed_time(e))
    out[name] = {"ms_per_call": dt, "kernels_ms": {nm: float(np.sum(x)) / n for nm, x in dur.items()}, "stats": dict(ops.LAST_FILTER_STATS)}
    res[name] = r
out["identical"] = bool(torch.equal(res["bf16_filter"][0], res["int8_cascade"][0]) and torch.equal(res["bf16_filter"][1], res["int8_cascade"][1]))
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe_cascade.json", "w"), indent=1)
