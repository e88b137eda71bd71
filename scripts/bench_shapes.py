"""A sweep over model variants on MovieLens-20M-shaped (Zipf) data, looking for performance cliffs: each line is one
configuration timed through the public API on one MI355X (seconds per epoch / per call).  Not a bench line; its output
feeds DESIGN.md's "what is slow and why" list."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import tensorrec_amd as T
from tensorrec_amd import loss_graphs as L, prediction_graphs as P, representation_graphs as R
from bench_configs import with_side_features, zipf_interactions

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_USERS, N_ITEMS = 138_493, 26_744


def timed_fit(name, model, inter, uf, itf, epochs=3, **kw):
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=1, **kw)
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=1, **kw)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    model.fit_partial(inter, uf, itf, epochs=epochs, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"case": name, "sec_per_epoch": (dt - one) / (epochs - 1), "first_call_sec": first, "one_epoch_call_sec": one}
    print(json.dumps(out), flush=True)
    return out


def timed(name, fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out = {"case": name, "sec_per_call": (time.perf_counter() - t0) / reps}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    which = set(sys.argv[1:])
    on = lambda k: not which or k in which     # noqa: E731
    uf = with_side_features(N_USERS, 30, 3, 1)
    itf = with_side_features(N_ITEMS, 20, 2, 2)
    inter = zipf_interactions(N_USERS, N_ITEMS, 160, 0)
    res = []
    if on("a"):
        m = T.TensorRec(n_components=128, loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("a: Linear d=128, dot, WMRB S=100", m, inter, uf, itf, n_sampled_items=100))
        if on("e"):
            sub = slice(0, 8192)
            res.append(timed("e1: predict_rank 8192 users x all items", lambda: m.predict_rank(uf[sub], itf), 1))
            res.append(timed("e2: predict 8192 users x all items", lambda: m.predict(uf[sub], itf), 1))
            res.append(timed("e3: predict_top_k all users, k=10", lambda: m.predict_top_k(uf, itf, k=10), 1))
            res.append(timed("e4: predict_rank_of_interactions, all 20M", lambda: m.predict_rank_of_interactions(uf, itf, inter), 1))
    if on("b"):
        m = T.TensorRec(n_components=128, seed=0)
        res.append(timed_fit("b: Linear d=128, dot, RMSE", m, inter, uf, itf))
    if on("c"):
        m = T.TensorRec(n_components=64, prediction_graph=P.CosineSimilarityPredictionGraph(), loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("c: Linear d=64, cosine, WMRB S=100", m, inter, uf, itf, n_sampled_items=100))
    if on("d"):
        m = T.TensorRec(n_components=64, n_tastes=3, loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("d: 3 tastes, Linear d=64, dot, WMRB S=100", m, inter, uf, itf, n_sampled_items=100))
    if on("g"):
        m = T.TensorRec(n_components=128, loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("g: as a, user_batch_size=32768", m, inter, uf, itf, n_sampled_items=100, user_batch_size=32768))
    if on("h"):
        m = T.TensorRec(n_components=128, loss_graph=L.BalancedWMRBLossGraph(), seed=0)
        res.append(timed_fit("h: Linear d=128, dot, BalancedWMRB S=100", m, inter, uf, itf, n_sampled_items=100))
    if on("i"):
        m = T.TensorRec(n_components=128, user_repr_graph=R.NormalizedLinearRepresentationGraph(),
                        item_repr_graph=R.NormalizedLinearRepresentationGraph(), loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("i: NormalizedLinear d=128, dot, WMRB S=100", m, inter, uf, itf, n_sampled_items=100))
    if on("j"):
        ns, ni = 20000, 5000
        m = T.TensorRec(n_components=64, loss_graph=L.SeparationDenseLossGraph(), seed=0)
        res.append(timed_fit("j: 20000 x 5000, Linear d=64, SeparationDense", m, inter[:ns, :ni], uf[:ns], itf[:ni]))
        m = T.TensorRec(n_components=64, loss_graph=L.RMSEDenseLossGraph(), seed=0)
        res.append(timed_fit("j2: 20000 x 5000, Linear d=64, RMSEDense", m, inter[:ns, :ni], uf[:ns], itf[:ni]))
    for dd in (10, 50, 100):
        if on("k%d" % dd):
            m = T.TensorRec(n_components=dd, loss_graph=L.WMRBLossGraph(), seed=0)
            res.append(timed_fit("k%d: Linear d=%d, dot, WMRB S=100" % (dd, dd), m, inter, uf, itf, n_sampled_items=100))
            m = T.TensorRec(n_components=dd, prediction_graph=P.EuclideanSimilarityPredictionGraph(),
                            loss_graph=L.WMRBLossGraph(), seed=0)
            res.append(timed_fit("k%d-euclid: Linear d=%d, euclidean, WMRB S=100" % (dd, dd), m, inter, uf, itf,
                                 n_sampled_items=100))
            res.append(timed("k%d: predict_top_k all users" % dd, lambda: m.predict_top_k(uf, itf, k=10), 1))
    if on("m"):
        # the README / generate_dummy_data defaults: hashed features, 20 per row, 200 columns
        di, du, dit = T.util.generate_dummy_data(num_users=15000, num_items=30000, interaction_density=.00045,
                                                 random_state=0)
        m = T.TensorRec(n_components=100, seed=0)
        res.append(timed_fit("m1: README defaults 15000 x 30000, 200 features, d=100, RMSE", m, di, du, dit, epochs=5))
        res.append(timed("m2: predict_rank 15000 x 30000 (to host)", lambda: m.predict_rank(du, dit), 1))
        res.append(timed("m3: predict_similar_items, 100 items, n_similar=10",
                         lambda: m.predict_similar_items(dit, item_ids=list(range(100)), n_similar=10), 1))
        pr = m.predict_rank(du, dit)
        res.append(timed("m4: recall/precision/ndcg @10 from the rank matrix (host)",
                         lambda: (T.eval.recall_at_k(pr, di, 10), T.eval.precision_at_k(pr, di, 10), T.eval.ndcg_at_k(pr, di, 10)), 1))
        m = T.TensorRec(n_components=100, loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("m5: same data, WMRB S=1000", m, di, du, dit, epochs=5, n_sampled_items=1000))
    if on("n"):
        m = T.TensorRec(n_components=64, n_tastes=3, attention_graph=R.LinearRepresentationGraph(),
                        loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("n: 3 tastes + linear attention, d=64, WMRB S=100", m, inter, uf, itf, n_sampled_items=100))
        bag = sp.random(N_ITEMS, 5000, density=0.06, random_state=3, dtype=np.float32, format="csr")      # 300 per row
        m = T.TensorRec(n_components=128, loss_graph=L.WMRBLossGraph(), seed=0)
        res.append(timed_fit("n2: item features = 5000-column bag, 300 per row, d=128, WMRB", m, inter, uf, bag,
                             n_sampled_items=100))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bench_shapes.json"), "w"), indent=1)
