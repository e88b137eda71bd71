"""BASELINE.json configs[4] (MovieLens-20M-shaped: 138,493 x 26,744, identity (+) 1,148 indicator columns, ReLU d=256 + Euclidean +
WMRB, S = 2,674) -- the fit loop alone, for rocprofv3 (--kernel-trace --stats, and the PMC passes of scripts/gpu_profile.sh):
python scripts/profile_cfg4.py [epochs]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_records as BR
import tensorrec_amd as T
from tensorrec_amd.representation_graphs import ReLURepresentationGraph
from tensorrec_amd.prediction_graphs import EuclideanSimilarityPredictionGraph

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_users, n_items, per_user, d = 138_493, 26_744, 160, 256
rng = np.random.default_rng(1)
S = n_items // 10
inter = BR._zipf_interactions(n_users, n_items, per_user, rng, exponent=0.8)
import scipy.sparse as sp
uf = sp.identity(n_users, dtype=np.float32, format="csr")
itf = BR._side_features(n_items, 1148, 8, rng)
model = T.TensorRec(n_components=d, user_repr_graph=ReLURepresentationGraph(), item_repr_graph=ReLURepresentationGraph(),
                    prediction_graph=EuclideanSimilarityPredictionGraph(), loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
model.fit_partial(inter, uf, itf, epochs=1, learning_rate=0.01, n_sampled_items=S)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.fit_partial(inter, uf, itf, epochs=epochs, learning_rate=0.01, n_sampled_items=S)
torch.cuda.synchronize()
print("cfg4: %.4f s per epoch over %d epochs (incl. the per-call upload check); %d sampled pairs + %d interactions per epoch"
      % ((time.perf_counter() - t0) / epochs, epochs, n_users * S, inter.nnz))
