"""Where the one-pass WMRB step's time goes: trec_wmrb_fused_step alone on the bench's fit workload (1M users x 1M items,
d = 128, 20 interactions + 100 samples per user) with parts switched off (tuning knob wmrb_ablate; results are wrong
in ablated runs, only the time is read).
Needs a library built with the ablation branches: make -C tensorrec_amd/csrc FLAGS_wmrb_fused=-DTREC_WMRB_ABLATE (the shipped
build has none of them and ignores the knob)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops, _native as N
from tensorrec_amd.sparse import Interactions
U = I = int(os.environ.get("N", 1_000_000)); d, per_user, S = 128, 20, 100
rng = np.random.default_rng(0)
cols = rng.integers(0, I, size=(U, per_user), dtype=np.int32)
m = sp.csr_matrix((np.ones(U * per_user, np.float32), cols.reshape(-1), np.arange(0, (U + 1) * per_user, per_user, dtype=np.int64)), shape=(U, I))
m.sum_duplicates(); m.data[:] = 1.0
inter = Interactions(m, U, I, "cuda")
u = torch.randn((U, d), device="cuda") * 0.1; v = torch.randn((I, d), device="cuda") * 0.1
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
samples = ops.sample_items(U, I, S, False, 0, 1)
out = {}
modes = (("full", 0), ("no histogram atomics", 1), ("no loss phases", 2), ("no dU", 4), ("gathers + dots only", 7))
if os.environ.get("ONLY_FULL"):           # A/B of library builds (TREC_HIP_LIB): the shipped kernel's time and a checksum of its outputs
    modes = modes[:1]
for name, ab in modes:
    N.set_tuning("wmrb_ablate", ab)
    ts = []
    for it in range(4):
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.KERNEL_EVENTS = []
        res = ops.wmrb_fused_step(u, v, ub, ib, inter, samples)
        torch.cuda.synchronize()
        ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        ts.append([e0.elapsed_time(e1) for n, e0, e1 in ev if n == "wmrb_fused_step"][0])
    out[name] = round(float(np.mean(ts[1:])), 3)
    print(name, out[name], "ms", flush=True)
N.set_tuning("wmrb_ablate", 0)
if os.environ.get("ONLY_FULL"):
    import hashlib
    out["checksum"] = hashlib.sha1(b"".join(t.detach().cpu().numpy().tobytes() for t in res[:3])).hexdigest()[:16]
    print("library", os.environ.get("TREC_HIP_LIB", "default"), "full", out["full"], "ms  checksum", out["checksum"], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fused_ablate.json"), "w"), indent=1)
