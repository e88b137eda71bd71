"""Randomised stress of the exact top-k paths: for many shapes / data distributions the int8 -> bf16 -> fp32 cascade and
the bf16 filter must return exactly what the all-fp32 MFMA path returns (values and ids)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops

rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
n_cases = int(os.environ.get("CASES", 40))
g = torch.Generator(device="cuda"); g.manual_seed(int(os.environ.get("SEED", 1)))
kinds = ["gauss", "normalised", "heavy_tail", "sparse", "integers", "clustered", "scaled_rows"]
res = []
t0 = time.time()
for c in range(n_cases):
    d = int(rng.choice([64, 100, 128, 40]))
    k = int(rng.choice([1, 5, 10, 12, 16]))
    n_u = int(rng.integers(1, 5000))
    n_i = int(rng.choice([rng.integers(20_000, 60_000), rng.integers(262_144, 700_000)]))
    kind = kinds[c % len(kinds)]
    def make(n):
        x = torch.randn((n, d), device="cuda", generator=g)
        if kind == "normalised": x = ops.l2_normalize_rows(x)
        elif kind == "heavy_tail": x = x * torch.exp(1.5 * torch.randn((n, d), device="cuda", generator=g))
        elif kind == "sparse": x = x * (torch.rand((n, d), device="cuda", generator=g) < 0.1)
        elif kind == "integers": x = torch.round(x * 2)
        elif kind == "clustered": x = torch.randn((8, d), device="cuda", generator=g)[torch.randint(0, 8, (n,), device="cuda", generator=g)] + 0.05 * x
        elif kind == "scaled_rows": x = x * torch.exp(2.0 * torch.randn((n, 1), device="cuda", generator=g))
        return x.contiguous()
    u, v = make(n_u), make(n_i)
    biased = bool(rng.integers(0, 2))
    scale = float(v.abs().mean() * u.abs().mean() * d ** 0.5)
    ub = (torch.randn(n_u, device="cuda", generator=g) * scale * 0.3) if biased else None
    ib = (torch.randn(n_i, device="cuda", generator=g) * scale * 0.3) if biased else None
    uop = ops.score_prep_filter(u); iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
    ev, ei = ops.score_topk(uop.f32, iop.f32, ops.DTYPE_F32, uop.kpad, k, ub, ib, ops.MODE_DOT, method="two_stage")
    out = {"case": c, "kind": kind, "d": d, "k": k, "users": n_u, "items": n_i, "biased": biased}
    for name, pre in (("bf16_filter", None), ("cascade", "int8")):
        uo = ops.score_prep_filter(u, sort_users=True, k=k) if pre == "int8" else uop
        fv, fi = ops.score_topk_filtered(uo, iop, k, ub, ib, prefilter=pre)
        ok = bool(torch.equal(fi, ei) and torch.equal(fv, ev))
        out[name] = ok
        out[name + "_stats"] = {k2: v2 for k2, v2 in ops.LAST_FILTER_STATS.items() if k2 in ("prefilter", "flagged_users", "refined_rows")}
    res.append(out)
    print(json.dumps(out))
bad = [r for r in res if not (r["bf16_filter"] and r["cascade"])]
summary = {"cases": len(res), "failures": len(bad), "seconds": time.time() - t0,
           "cascade_ran": sum(1 for r in res if r["cascade_stats"].get("prefilter") == "int8"),
           "cascade_fell_back": sum(1 for r in res if str(r["cascade_stats"].get("prefilter", "")).startswith("int8 (")),
           "not_offered": sum(1 for r in res if "prefilter" not in r["cascade_stats"])}
print(json.dumps(summary))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"summary": summary, "cases": res}, open("gpurun_out/fuzz_cascade.json", "w"), indent=1)
sys.exit(1 if bad else 0)
