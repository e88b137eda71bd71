"""BASELINE.json configs[3] and configs[4] on ONE MI355X, through the public API (these are parity-test shapes, not the
bench line; this script only puts measured single-GPU numbers next to them in DESIGN.md):

  configs[3]  10M items, d=128, CosineSimilarity, top-10 -- the whole item set on one GPU (an 8-GPU run scores 1.25M
              items per rank) for 131,072 users
  configs[4]  MovieLens-20M-shaped fit: 138,493 users x 26,744 items, 20M interactions, identity + indicator side
              features, ReLURepresentation d=256 + EuclideanSimilarity, WMRB, 100 sampled items per user
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import tensorrec_amd as T
from tensorrec_amd.representation_graphs import ReLURepresentationGraph
from tensorrec_amd.prediction_graphs import CosineSimilarityPredictionGraph, EuclideanSimilarityPredictionGraph
from tensorrec_amd.loss_graphs import WMRBLossGraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def with_side_features(n, n_side, per_row, seed):
    """identity block | `per_row` random indicator columns out of `n_side` (genres / demographics)"""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n_side, size=(n, per_row), dtype=np.int64) + n
    ident = np.arange(n, dtype=np.int64)[:, None]
    allc = np.concatenate([ident, cols], axis=1).reshape(-1)
    indptr = np.arange(0, (n + 1) * (per_row + 1), per_row + 1, dtype=np.int64)
    m = sp.csr_matrix((np.ones(allc.size, np.float32), allc, indptr), shape=(n, n + n_side))
    m.sum_duplicates()
    m.data[:] = 1.0
    return m


def zipf_interactions(n_users, n_items, per_user, seed):
    rng = np.random.default_rng(seed)
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.8
    pop /= pop.sum()
    cols = rng.choice(n_items, size=(n_users, per_user), p=pop).astype(np.int32)
    indptr = np.arange(0, (n_users + 1) * per_user, per_user, dtype=np.int64)
    m = sp.csr_matrix((np.ones(n_users * per_user, np.float32), cols.reshape(-1), indptr), shape=(n_users, n_items))
    m.sum_duplicates()
    m.data[:] = 1.0
    return m


def config3(n_items=10_000_000, n_users=131_072, d=128, k=10, reps=3):
    uf = sp.identity(n_users, dtype=np.float32, format="csr")
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    model = T.TensorRec(n_components=d, prediction_graph=CosineSimilarityPredictionGraph(), precision="bf16", seed=0)
    # weights only: a one-interaction fit builds the tables (the timed call is predict_top_k)
    inter = sp.csr_matrix((np.ones(1, np.float32), ([0], [0])), shape=(n_users, n_items))
    model.fit_partial(inter, uf, itf, epochs=1)
    model.predict_top_k(uf, itf, k=k, return_device=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        vals, idx = model.predict_top_k(uf, itf, k=k, return_device=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # size-independent properties at the full shape: lists sorted, ids in range and distinct, cosine scores in [-1, 1]
    v, i = vals.cpu().numpy(), idx.cpu().numpy()
    assert (np.diff(v, axis=1) <= 0).all() and (i >= 0).all() and (i < n_items).all()
    assert all(len(set(r)) == k for r in i[:2048]) and np.abs(v).max() <= 1.0 + 1e-2
    return {"case": "configs[3] on one GPU: %d users x %d items, d=%d, cosine, top-%d (features -> lists, host API)"
            % (n_users, n_items, d, k), "sec_per_call": dt, "predictions_per_sec": n_users * n_items / dt}


def config4(n_users=138_493, n_items=26_744, per_user=160, d=256, S=100, epochs=4):
    uf = with_side_features(n_users, 30, 3, 1)
    itf = with_side_features(n_items, 20, 2, 2)
    inter = zipf_interactions(n_users, n_items, per_user, 0)
    model = T.TensorRec(n_components=d, user_repr_graph=ReLURepresentationGraph(),
                        item_repr_graph=ReLURepresentationGraph(),
                        prediction_graph=EuclideanSimilarityPredictionGraph(), loss_graph=WMRBLossGraph(), seed=0)
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=1, n_sampled_items=S)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=1, n_sampled_items=S)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=epochs, n_sampled_items=S)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_epoch = (dt - one) / (epochs - 1)
    return {"case": "configs[4] on one GPU: ML-20M-shaped, ReLU d=%d + Euclidean, WMRB, %d samples/user" % (d, S),
            "users": n_users, "items": n_items, "interactions": int(inter.nnz), "sec_per_epoch": per_epoch,
            "fit_epochs_per_sec": 1.0 / per_epoch, "first_call_sec": first, "call_overhead_sec": one - per_epoch,
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4"]
    res = []
    if "4" in which:
        res.append(config4()); print(json.dumps(res[-1]), flush=True)
    if "3" in which:
        res.append(config3()); print(json.dumps(res[-1]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
