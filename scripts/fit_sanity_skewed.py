"""Training-quality sanity on skewed data (not a test: a minute of GPU): Zipf-popular items + planted taste clusters, side
features that reveal the cluster, ReLU + Euclidean + WMRB and Linear + dot + WMRB.  The chunked gathers, the structured
Euclidean gradient and the split-K GEMM all sit on this path: the loss must fall and held-out positives must rank far
above chance."""
import sys, os, json, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tensorrec_amd as T
from tensorrec_amd import loss_graphs as L, prediction_graphs as P, representation_graphs as R

rng = np.random.default_rng(0)
nu, ni, nc, per = 30000, 6000, 20, 40
ucl, icl = rng.integers(0, nc, nu), rng.integers(0, nc, ni)
pop = 1.0 / np.arange(1, ni + 1) ** 0.9
rows, cols = [], []
for c in range(nc):
    us = np.where(ucl == c)[0]
    p = pop * np.where(icl == c, 12.0, 1.0)          # a user's cluster is 12x as likely, on top of popularity
    p /= p.sum()
    ch = rng.choice(ni, size=(len(us), per), p=p)
    rows.append(np.repeat(us, per)); cols.append(ch.reshape(-1))
m = sp.csr_matrix((np.ones(nu * per, np.float32), (np.concatenate(rows), np.concatenate(cols))), shape=(nu, ni))
m.sum_duplicates(); m.data[:] = 1.0
coo = m.tocoo()
test_mask = rng.random(coo.nnz) < 0.1
train = sp.csr_matrix((coo.data[~test_mask], (coo.row[~test_mask], coo.col[~test_mask])), shape=m.shape)
test = sp.csr_matrix((coo.data[test_mask], (coo.row[test_mask], coo.col[test_mask])), shape=m.shape)
print("interactions", train.nnz, "most popular item", int(np.bincount(train.indices).max()), flush=True)


def feats(n, cl):
    ident = sp.identity(n, dtype=np.float32, format="csr")
    onehot = sp.csr_matrix((np.ones(n, np.float32), (np.arange(n), cl)), shape=(n, nc))
    return sp.hstack([ident, onehot], format="csr")


uf, itf = feats(nu, ucl), feats(ni, icl)
out = []
for name, kw in (("Linear d=32, dot, WMRB", dict()),
                 ("ReLU d=32, euclidean, WMRB", dict(user_repr_graph=R.ReLURepresentationGraph(), item_repr_graph=R.ReLURepresentationGraph(),
                                                     prediction_graph=P.EuclideanSimilarityPredictionGraph())),
                 ("Linear d=30 (not a multiple of 4), cosine, BalancedWMRB", dict(n_components=30, prediction_graph=P.CosineSimilarityPredictionGraph(),
                                                                                  loss_graph=L.BalancedWMRBLossGraph()))):
    args = dict(n_components=32, loss_graph=L.WMRBLossGraph(), seed=1)
    args.update(kw)
    model = T.TensorRec(**args)
    losses = []
    for ep in range(4):
        model._capture = {}
        model.fit_partial(train, uf, itf, epochs=10, learning_rate=0.01, n_sampled_items=100)
        losses.append(float(model._capture['loss'].mean()))
    model._capture = None
    pr = model.predict_rank_of_interactions(uf, itf, test)
    recall = float(np.nanmean(T.eval.recall_at_k(pr, test, k=60)))
    w = model.get_weights()
    assert all(np.isfinite(v).all() for v in w.values())
    rec = {"model": name, "loss_every_10_epochs": [round(l, 4) for l in losses], "heldout_recall@60": round(recall, 4),
           "chance": round(60 / ni, 4)}
    print(json.dumps(rec), flush=True)
    assert losses[-1] < losses[0] and recall > 5 * 60 / ni, rec
    out.append(rec)
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "fit_sanity_skewed.json"), "w"), indent=1)
