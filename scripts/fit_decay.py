"""The fit of bench.py (1M x 1M, d = 128, WMRB, 20 interactions + 100 samples per user) epoch by epoch: ms per one-epoch call and
the fraction of the sampled pairs whose coefficient is not 0 (the pairs the item side still sorts and gathers) as the model learns.
python scripts/fit_decay.py [calls] [clusters=C] [alpha=A] [tuning=value ...]
alpha=A (default 1e-5, the API's): the reference adds alpha * l2 to EVERY element of the loss vector before the optimiser sums it
(tensorrec.py:487-489), so the penalty's weight grows with the number of positive interactions -- 2e7 here, 200 * |w|^2 / 2 at the
default: the weights are held near 0 for dozens of epochs whatever the data.  alpha=0 shows what the loss alone does.
clusters=C (default 0 = bench.py's uniform-random interactions, which no model can learn): user u and item i belong to cluster u % C /
i % C and every user's 20 interactions are items of its own cluster -- structure a factorisation learns within a few epochs."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
import tensorrec_amd as T
from tensorrec_amd import ops, _native as N

if __name__ == "__main__":
    args = sys.argv[1:]
    calls = int(args[0]) if args and "=" not in args[0] else 30
    clusters, alpha = 0, 0.00001
    for t in [a for a in args if "=" in a]:
        name, value = t.split("=")
        if name == "clusters":
            clusters = int(value)
        elif name == "alpha":
            alpha = float(value)
        else:
            N.set_tuning(name, int(value))
    n_users = n_items = int(os.environ.get("FIT_DECAY_N", 1_000_000))
    rng = np.random.default_rng(1000)
    if clusters:
        per = n_items // clusters
        cols = (rng.integers(0, per, size=(n_users, 20), dtype=np.int32) * clusters
                + (np.arange(n_users, dtype=np.int32) % clusters)[:, None]).astype(np.int32)
    else:
        cols = rng.integers(0, n_items, size=(n_users, 20), dtype=np.int32)
    inter = sp.csr_matrix((np.ones(n_users * 20, np.float32), cols.reshape(-1), np.arange(0, (n_users + 1) * 20, 20, dtype=np.int64)),
                          shape=(n_users, n_items))
    inter.sum_duplicates()
    inter.data[:] = 1.0
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    model = T.TensorRec(n_components=128, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
    out = []
    for c in range(calls):
        ops.KERNEL_EVENTS = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit_partial(inter, uf, itf, epochs=1, alpha=alpha, n_sampled_items=100)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        dur = {}
        for name, s_, e_ in events:
            dur[name] = dur.get(name, 0.0) + s_.elapsed_time(e_)
        st = ops.LAST_FUSED_STATS
        kept = int(st["sampled_pairs_kept"].item()) / st["sampled_pairs"] if st.get("sampled_pairs") else None
        out.append({"call": c, "call_ms_with_upload": round(ms, 2), "kept_fraction_of_sampled_pairs": kept,
                    "coefficient_0_fraction": int(st["sampled_pairs_with_coefficient_0"].item()) / st["sampled_pairs"],
                    "loss_mean": float(st["loss_mean"].item()),
                    "kernels_ms": {k: round(v, 3) for k, v in dur.items()}})
        print(json.dumps(out[-1]), flush=True)
