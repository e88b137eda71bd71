"""K1 on 1M rows x 20 non-zeros over 1M feature columns, d = 128 (the weight table does not fit any cache): the launch
profiled by scripts/gpu_pmc_cmd.sh for bench.py's roofline_k1_multi_nnz traffic figure."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops
from tensorrec_amd.sparse import SparseFeatures
n_rows, nnz_row, F, d = 1_000_000, 20, 1_000_000, 128
rng = np.random.default_rng(5)
cols = rng.integers(0, F, size=(n_rows, nnz_row), dtype=np.int32)
cols.sort(axis=1)
m = sp.csr_matrix((rng.random(n_rows * nnz_row, dtype=np.float32), cols.reshape(-1),
                   np.arange(0, (n_rows + 1) * nnz_row, nnz_row, dtype=np.int64)), shape=(n_rows, F))
f = SparseFeatures(m, "cuda")
w = torch.randn((F, d), device="cuda")
for _ in range(4):
    ops.spmm_raw(f.indptr, f.indices, f.values, None, n_rows, f.nnz, w)
torch.cuda.synchronize()
