"""A/B of tuning knobs on FITTED weights at the bench shape: fit once (WMRB, planted-cluster Zipf interactions, 20 epochs), then
TensorRec.predict_top_k(k=10) under every knob setting given on the command line ("name=value[,name=value]" per setting): ms per
call, per-kernel HIP-event times, the cascade's statistics, and whether every setting returns the same lists.
python scripts/trained_ab.py cascade_prerefine=0 cascade_prerefine=1"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import tensorrec_amd as T
from tensorrec_amd import ops, _native as N
from tensorrec_amd.synth import planted_cluster_interactions

U = I = int(os.environ.get("N", 1_000_000)); d = 128
inter, _, _, _ = planted_cluster_interactions(U, I, 256, 20, seed=0, holdout=0.05)
uf = sp.identity(U, dtype=np.float32, format="csr"); itf = sp.identity(I, dtype=np.float32, format="csr")
model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
model.fit_partial(inter, uf, itf, epochs=int(os.environ.get("EPOCHS", 20)), learning_rate=0.1, n_sampled_items=100)
torch.cuda.synchronize()
ref = None
for setting in sys.argv[1:] or ["cascade_prerefine=1"]:
    knobs = dict(kv.split("=") for kv in setting.split(","))
    for k_, v_ in knobs.items():
        N.set_tuning(k_, int(v_))
    model.predict_top_k(uf, itf, k=10, return_device=True)
    torch.cuda.synchronize()
    ops.KERNEL_EVENTS = []
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        v, i = model.predict_top_k(uf, itf, k=10, return_device=True)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dur = {}
    for n_, a, b in ev:
        dur.setdefault(n_, []).append(a.elapsed_time(b))
    same = None
    if ref is None:
        ref = (v.clone(), i.clone())
    else:
        same = bool(torch.equal(v, ref[0]) and torch.equal(i, ref[1]))
    st = {k_: v_ for k_, v_ in ops.LAST_FILTER_STATS.items() if k_ in ("refined_rows", "flagged_users", "prerefined_pairs", "prefilter")}
    print(setting, "ms min %.2f" % min(ts), {n_.replace("score_gemm_", "").replace("topk_", ""): round(float(np.sum(x)) / 4, 2) for n_, x in dur.items()},
          st, "same lists as the first setting:", same, flush=True)
