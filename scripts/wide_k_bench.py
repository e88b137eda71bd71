"""17 <= k <= 64 through the cascade (ops.score_topk_filtered_wide) at 200k users x 1M items, d = 128, biased: ms per call, the
bracketed launch groups, and exact equality with the all-fp32 path on a sample of users."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops
from tensorrec_amd.ops_base import DTYPE_F32

U = int(os.environ.get("U", 200_000)); I = int(os.environ.get("I", 1_000_000)); d = int(os.environ.get("D", 128))
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.randn((U, d), device="cuda", generator=g) * 0.1
v = torch.randn((I, d), device="cuda", generator=g) * 0.1
ub = torch.randn(U, device="cuda", generator=g) * 0.01
ib = torch.randn(I, device="cuda", generator=g) * 0.01
out = {"users": U, "items": I, "d": d}
from tensorrec_amd import _native as N
if os.environ.get("WIDE_PREREFINE") is not None:            # A/B: WIDE_PREREFINE=0 -> the wide route without the pre-refinement
    N.set_tuning("wide_prerefine", int(os.environ["WIDE_PREREFINE"]))
    out["wide_prerefine"] = int(os.environ["WIDE_PREREFINE"])
if os.environ.get("EARLY_DENSE") is not None:
    N.set_tuning("prerefine_early_dense", int(os.environ["EARLY_DENSE"]))
    out["prerefine_early_dense"] = int(os.environ["EARLY_DENSE"])
for k in tuple(int(x) for x in os.environ.get("KS", "32,64").split(",")):
    def step():
        uop = ops.score_prep_filter(u, sort_users=True, k=k, user_bias=ub)
        iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
        return ops.score_topk_filtered_wide(uop, iop, k, ub, ib)
    step(); torch.cuda.synchronize()
    ops.KERNEL_EVENTS = []
    n = 3
    t0 = time.perf_counter()
    for _ in range(n): r = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dur = {}
    for nm, s, e in ev: dur.setdefault(nm, []).append(s.elapsed_time(e))
    # the all-fp32 answer on a sample of users
    sel = torch.arange(0, U, max(1, U // 512), device="cuda")
    uo = ops.score_prep(u[sel].contiguous(), DTYPE_F32); io = ops.score_prep(v, DTYPE_F32)
    slab = ops.score_store(uo[0], io[0], DTYPE_F32, uo[2], ub[sel].contiguous(), ib)
    fv, fi = ops.topk_from_scores(slab, k)
    same = bool(torch.equal(fv, r[0][sel]) and torch.equal(fi, r[1][sel]))
    out["k%d" % k] = {"ms_per_call": dt, "groups_ms": {nm: float(np.sum(x)) / n for nm, x in dur.items()},
                      "stats": {kk: vv for kk, vv in dict(ops.LAST_FILTER_STATS).items()}, "equals_fp32_path_on_sample": same}
print(json.dumps(out, indent=1, default=str))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/wide_k_bench%s.json" % os.environ.get("TAG", ""), "w"), indent=1, default=str)
