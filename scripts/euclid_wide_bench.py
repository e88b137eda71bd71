"""Euclidean top-k for 13 <= k <= 48 through the wide cascade + certificate (ops.score_topk_euclid_filtered) at 200k users x 1M items,
d = 128, biased (sigma_b = 0.01): ms per call, users left without a certificate, and exact equality with fp32 score slabs
(trec_score_gemm_store in MODE_EUCLIDEAN + ops.topk_from_scores: the route these k took before) on a sample of users."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorrec_amd import ops
from tensorrec_amd.ops_base import DTYPE_F32, MODE_EUCLIDEAN

U = int(os.environ.get("U", 200_000)); I = int(os.environ.get("I", 1_000_000)); d = int(os.environ.get("D", 128))
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.randn((U, d), device="cuda", generator=g) * 0.1
v = torch.randn((I, d), device="cuda", generator=g) * 0.1
ub = torch.randn(U, device="cuda", generator=g) * 0.01
ib = torch.randn(I, device="cuda", generator=g) * 0.01
out = {"users": U, "items": I, "d": d, "item_bias_sigma": 0.01}
from tensorrec_amd import _native as N
for knob in ("euclid_lambda_sample", "euclid_second_pass_wider"):          # A/B: e.g. EUCLID_LAMBDA_SAMPLE=0 EUCLID_SECOND_PASS_WIDER=0
    if os.environ.get(knob.upper()) is not None:
        N.set_tuning(knob, int(os.environ[knob.upper()])); out[knob] = int(os.environ[knob.upper()])
for k in tuple(int(x) for x in os.environ.get("KS", "20,40").split(",")):
    step = lambda: ops.score_topk_euclid_filtered(u, v, k, ub, ib)
    step(); torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n): r = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    stats = dict(ops.LAST_FILTER_STATS)
    # the slab route on a sample of users (timed per user batch of 256: what the whole call would cost that way)
    sel = torch.arange(0, U, max(1, U // 256), device="cuda")[:256]
    uo = ops.score_prep(u[sel].contiguous(), DTYPE_F32, want_sqnorm=True); io = ops.score_prep(v, DTYPE_F32, want_sqnorm=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    slab = ops.score_store(uo[0], io[0], DTYPE_F32, uo[2], ub[sel].contiguous(), ib, MODE_EUCLIDEAN, uo[1], io[1])
    fv, fi = ops.topk_from_scores(slab, k)
    torch.cuda.synchronize(); slab_ms = (time.perf_counter() - t0) * 1e3
    same = bool(torch.equal(fv, r[0][sel]) and torch.equal(fi, r[1][sel]))
    out["k%d" % k] = {"ms_per_call": dt, "stats": {a: b for a, b in stats.items()}, "equals_fp32_slabs_on_sample": same,
                      "slab_route_ms_per_256_users": slab_ms, "slab_route_ms_extrapolated": slab_ms * U / 256.0}
print(json.dumps(out, indent=1, default=str))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/euclid_wide_bench%s.json" % os.environ.get("TAG", ""), "w"), indent=1, default=str)
