"""K1 micro-benchmark: CSR gather-SpMM variants on the BASELINE shapes (HIP events on the launching stream)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops, _native as N
from tensorrec_amd.sparse import SparseFeatures

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def case(name, feats, d, results):
    w = torch.randn((feats.shape[1], d), device="cuda")
    out = torch.empty((feats.shape[0], d), device="cuda")
    nnz, rows = feats.nnz, feats.shape[0]
    alg = nnz * 8 + (rows + 1) * 8 + nnz * d * 4 + rows * d * 4
    for r in (0, 1):
        for nt in (0, 1):
            N.set_tuning("spmm_ntload", r); N.set_tuning("spmm_nt", nt)
            ms = timeit(lambda: ops.spmm_raw(feats.indptr, feats.indices, feats.values, None, rows, nnz, w, out=out))
            results.append({"case": name, "nontemporal_load": r, "nontemporal_store": nt, "ms": ms,
                            "GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000})
            print(results[-1], flush=True)

res = []
n = 1_000_000
case("identity 1M x 1M, d=128", SparseFeatures(sp.identity(n, dtype=np.float32, format="csr"), "cuda"), 128, res)
rng = np.random.default_rng(0)
perm = rng.permutation(n)
case("permutation 1M x 1M, d=128", SparseFeatures(sp.csr_matrix((np.ones(n, np.float32), (np.arange(n), perm)), shape=(n, n)), "cuda"), 128, res)
case("identity 1M, d=64", SparseFeatures(sp.identity(n, dtype=np.float32, format="csr"), "cuda"), 64, res)
m = sp.random(138493, 27892, density=20 / 27892, random_state=0, dtype=np.float32, format="csr")
case("ML-20M-like users 138k x 27.9k feats, ~20 nnz/row, d=256", SparseFeatures(m, "cuda"), 256, res)
N.set_tuning("spmm_ntload", -1); N.set_tuning("spmm_nt", 1)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_k1.json"), "w"), indent=1)
