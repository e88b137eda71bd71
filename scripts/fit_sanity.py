"""Large-scale sanity of the fit path (not a test: minutes of GPU): the one-pass WMRB step and the composed kernels give the\nsame loss trajectory on 300k x 300k, the loss falls, the weights stay finite."""
import sys, os, numpy as np, scipy.sparse as sp, torch, logging
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tensorrec_amd as T
n, d, S = 300_000, 128, 100
rng = np.random.default_rng(0)
# planted structure: user u likes items near (u * 7) % n  -> the loss must fall and positives must outrank samples
cols = ((np.arange(n)[:, None] * 7 + rng.integers(0, 50, size=(n, 20))) % n).astype(np.int32)
inter = sp.csr_matrix((np.ones(n * 20, np.float32), cols.reshape(-1), np.arange(0, (n + 1) * 20, 20, dtype=np.int64)), shape=(n, n))
inter.sum_duplicates(); inter.data[:] = 1
uf = sp.identity(n, dtype=np.float32, format="csr"); itf = sp.identity(n, dtype=np.float32, format="csr")
res = {}
for fused in (1, 0):
    T._native.set_tuning("wmrb_fused", fused)
    m = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
    losses = []
    for ep in range(6):
        m._capture = {}
        m.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.05, n_sampled_items=S)
        losses.append(float(m._capture['loss'].mean()))
    m._capture = None
    res[fused] = losses
    print("fused", fused, ["%.4f" % l for l in losses], flush=True)
    w = m.get_weights()
    assert all(np.isfinite(v).all() for v in w.values())
assert all(b < a for a, b in zip(res[1], res[1][1:])) and np.allclose(res[1], res[0], rtol=1e-5), res
vals, idx = m.predict_top_k(uf[:2000], itf, k=10)
hit = np.mean([len(set(idx[u]) & set(cols[u])) > 0 for u in range(2000)])
print("users with a positive in their top-10 after 6 epochs: %.2f" % hit)
