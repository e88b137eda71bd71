import sys, time, numpy as np, scipy.sparse as sp, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import tensorrec_amd as T
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_fit import zipf_interactions
n_users, n_items, d, S = 943, 1682, 64, 168
uf = sp.identity(n_users, dtype=np.float32, format="csr"); itf = sp.identity(n_items, dtype=np.float32, format="csr")
inter = zipf_interactions(n_users, n_items, 96, 0)
for graphs in (True, False):
    m = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0, hip_graphs=graphs)
    m.fit_partial(inter, uf, itf, epochs=5, n_sampled_items=S); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.fit_partial(inter, uf, itf, epochs=500, n_sampled_items=S); torch.cuda.synchronize()
    print("graphs", graphs, "ms/epoch %.3f" % ((time.perf_counter() - t0) / 500 * 1e3), flush=True)
for graphs in (True, False):
    m = T.TensorRec(n_components=d, seed=0, hip_graphs=graphs)         # RMSE, default model
    m.fit_partial(inter, uf, itf, epochs=5); torch.cuda.synchronize()
    t0 = time.perf_counter(); m.fit_partial(inter, uf, itf, epochs=500); torch.cuda.synchronize()
    print("rmse graphs", graphs, "ms/epoch %.3f" % ((time.perf_counter() - t0) / 500 * 1e3), flush=True)
