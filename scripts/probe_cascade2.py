"""Cascade experiments on one box: user clip levels, and the two halves of the users on two HIP streams (the memory-bound
table passes of one half under the MFMA-bound GEMM of the other)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops, _native as N

U = I = 1_000_000; d = 128; k = 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
out = {}

def one(us, ubs, iop):
    uop = ops.score_prep_filter(us)
    return ops.score_topk_filtered(uop, iop, k, ubs, ib, prefilter="int8")

def step_serial():
    iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    return one(u, ub, iop)

streams = [torch.cuda.Stream() for _ in range(4)]
def step_split(n_split):
    iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    # quantise the items once, on the main stream (a dummy user side fixes nothing: do it with the first slice's users)
    main = torch.cuda.current_stream()
    per = (U + n_split - 1) // n_split
    res = []
    # the item-side int8 operand is shared: prepare it with the first slice on the main stream, later slices only re-derive biases
    uop0 = ops.score_prep_filter(u[:per]); ops.score_prep_i8_pair(uop0, iop, ib)
    import copy
    for j in range(n_split):
        st = streams[j]
        st.wait_stream(main)
        with torch.cuda.stream(st):
            iopj = copy.copy(iop)
            iopj.scales = iop.scales.clone(); iopj.gstats8 = iop.gstats8.clone()
            r = one(u[j * per:(j + 1) * per], ub[j * per:(j + 1) * per], iopj)
            res.append(r)
    for st in streams[:n_split]: main.wait_stream(st)
    return torch.cat([r[0] for r in res]), torch.cat([r[1] for r in res])

def timeit(f, n=4):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

for clip in (50, 45, 40, 35):
    N.load().trec_set_tuning(b"i8_user_clip_x10", clip)
    ms, r = timeit(step_serial, 2)
    out["serial_clip%d" % clip] = {"ms": ms, "refined_rows": ops.LAST_FILTER_STATS.get("refined_rows"), "flagged": ops.LAST_FILTER_STATS.get("flagged_users")}
N.load().trec_set_tuning(b"i8_user_clip_x10", 50)
ms, ref = timeit(step_serial)
out["serial"] = ms
for ns in (2, 3, 4):
    ms, r = timeit(lambda: step_split(ns))
    out["split%d" % ns] = {"ms": ms, "identical": bool(torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]))}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/probe_cascade2.json", "w"), indent=1)
