"""BASELINE.json configs[1] (MovieLens-100K-shaped: 943 x 1,682, identity users, identity (+) 19 genre columns, Linear d = 64 + DotProduct +
WMRB, S = 168) -- the fit loop alone, for rocprofv3 --kernel-trace --stats: python scripts/profile_cfg1.py [epochs] [fit_step_coop 0/1]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
import bench_records as BR
import tensorrec_amd as T

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    T._native.set_tuning("fit_step_coop", int(sys.argv[2]))
rng = np.random.default_rng(0)
n_users, n_items, d, S, lr = 943, 1682, 64, 168, 0.05
inter = BR._zipf_interactions(n_users, n_items, 160, rng, exponent=1.0)
uf = sp.identity(n_users, dtype=np.float32, format="csr")
itf = BR._side_features(n_items, 19, 3, rng)
model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
model.fit_partial(inter, uf, itf, epochs=5, learning_rate=lr, n_sampled_items=S)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.fit_partial(inter, uf, itf, epochs=epochs, learning_rate=lr, n_sampled_items=S)
torch.cuda.synchronize()
print("cfg1: %.4f ms per epoch over %d epochs (incl. the per-call upload check)" % (1e3 * (time.perf_counter() - t0) / epochs, epochs))
