"""Latency of the exact top-k for SMALL user batches against a large catalogue (serving-sized calls): ms per call of
ops.score_topk_filtered (k = 10) and ops.score_topk_filtered_wide (k = 32) for 256 / 512 / 4,096 users x 1M items, d = 128, biased,
the item operand prepared once (it does not change between calls); bracketed launch groups of the 512-user case."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops

I, d = int(os.environ.get("I", 1_000_000)), 128
g = torch.Generator(device="cuda"); g.manual_seed(0)
v = torch.randn((I, d), device="cuda", generator=g) * 0.1
ib = torch.randn(I, device="cuda", generator=g) * 0.01
iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
out = {"items": I, "d": d}
if os.environ.get("MAX_CHUNKS"):                           # A/B: item chunks of the stage-1 launches at most
    from tensorrec_amd import _native as N
    N.set_tuning("cascade_max_chunks", int(os.environ["MAX_CHUNKS"])); out["cascade_max_chunks"] = int(os.environ["MAX_CHUNKS"])
for U in (256, 512, 4096):
    u = torch.randn((U, d), device="cuda", generator=g) * 0.1
    ub = torch.randn(U, device="cuda", generator=g) * 0.01
    for k in (10, 32):
        def step():
            uop = ops.score_prep_filter(u, sort_users=True, k=k, user_bias=ub)
            if k <= 16:
                return ops.score_topk_filtered(uop, iop, k, ub, ib, prefilter="int8")
            return ops.score_topk_filtered_wide(uop, iop, k, ub, ib)
        for _ in range(3): step()
        torch.cuda.synchronize()
        ops.KERNEL_EVENTS = [] if U == 512 else None
        n = 20
        t0 = time.perf_counter()
        for _ in range(n): r = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
        rec = {"ms_per_call": dt}
        if U == 512:
            ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            dur = {}
            for nm, s, e in ev: dur.setdefault(nm, []).append(s.elapsed_time(e))
            rec["groups_ms"] = {nm: float(np.sum(x)) / n for nm, x in dur.items()}
        out["users%d_k%d" % (U, k)] = rec
        print(U, k, rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/small_batch_bench%s.json" % os.environ.get("TAG", ""), "w"), indent=1)
