"""Device sampler (keyed Feistel permutation) vs the reference's host sampler (np.random.choice per user,
tensorrec/util.py:12-21) on the skewed sanity set of scripts/fit_sanity_skewed.py: same model, same seed, same epochs;
held-out recall@60 and the loss trajectory.  Result -> gpurun_out/sampler_ab.json (copied to profiles/)."""
import sys, os, json, time, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tensorrec_amd as T
from tensorrec_amd import loss_graphs as L

rng = np.random.default_rng(0)
nu, ni, nc, per = 30000, 6000, 20, 40
ucl, icl = rng.integers(0, nc, nu), rng.integers(0, nc, ni)
pop = 1.0 / np.arange(1, ni + 1) ** 0.9
rows, cols = [], []
for c in range(nc):
    us = np.where(ucl == c)[0]
    p = pop * np.where(icl == c, 12.0, 1.0)
    p /= p.sum()
    rows.append(np.repeat(us, per)); cols.append(rng.choice(ni, size=(len(us), per), p=p).reshape(-1))
m = sp.csr_matrix((np.ones(nu * per, np.float32), (np.concatenate(rows), np.concatenate(cols))), shape=(nu, ni))
m.sum_duplicates(); m.data[:] = 1.0
coo = m.tocoo()
test_mask = rng.random(coo.nnz) < 0.1
train = sp.csr_matrix((coo.data[~test_mask], (coo.row[~test_mask], coo.col[~test_mask])), shape=m.shape)
test = sp.csr_matrix((coo.data[test_mask], (coo.row[test_mask], coo.col[test_mask])), shape=m.shape)


def feats(n, cl):
    return sp.hstack([sp.identity(n, dtype=np.float32, format="csr"),
                      sp.csr_matrix((np.ones(n, np.float32), (np.arange(n), cl)), shape=(n, nc))], format="csr")


uf, itf = feats(nu, ucl), feats(ni, icl)
out = []
for name, make in (("device sampler (Feistel permutation, csrc/sampler.hip)", lambda s: T.DeviceSampler(s)),
                   ("host sampler (np.random.choice per user, util.py:12-21)", lambda s: T.HostSampler(np.random.RandomState(s)))):
    for seed in (1, 2, 3):
        model = T.TensorRec(n_components=32, loss_graph=L.WMRBLossGraph(), seed=seed, sampler=make(seed))
        losses, t0 = [], time.perf_counter()
        for ep in range(2):
            model._capture = {}
            model.fit_partial(train, uf, itf, epochs=10, learning_rate=0.01, n_sampled_items=100)
            losses.append(float(model._capture['loss'].mean()))
        dt = time.perf_counter() - t0
        model._capture = None
        pr = model.predict_rank_of_interactions(uf, itf, test)
        rec = {"sampler": name, "seed": seed, "loss_after_10_20_epochs": [round(l, 4) for l in losses],
               "heldout_recall@60": round(float(np.nanmean(T.eval.recall_at_k(pr, test, k=60))), 4),
               "heldout_ndcg@60": round(float(np.nanmean(T.eval.ndcg_at_k(pr, test, k=60))), 4),
               "fit_seconds_20_epochs": round(dt, 2)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "sampler_ab.json"), "w"), indent=1)
