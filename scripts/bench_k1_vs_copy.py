"""K1 on 1M identity rows, d=128 against the chip's own 1:1 copy rate: torch copy_ of the same 512 MB -> 512 MB, measured
in the same process (a gather that moves 1.04 GB cannot beat a straight copy of 1.02 GB by much)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops, _native as N
from tensorrec_amd.sparse import SparseFeatures


def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


n, d = 1_000_000, 128
f = SparseFeatures(sp.identity(n, dtype=np.float32, format="csr"), "cuda")
w = torch.randn((n, d), device="cuda")
out = torch.empty((n, d), device="cuda")
alg = f.nnz * 8 + (n + 1) * 8 + f.nnz * d * 4 + n * d * 4
for rep in range(2):
    ms = timeit(lambda: out.copy_(w))
    print("copy_ 512 MB -> 512 MB: %.4f ms = %.0f GB/s" % (ms, 2 * n * d * 4 / ms / 1e6), flush=True)
    ms = timeit(lambda: ops.spmm_raw(f.indptr, f.indices, f.values, None, n, f.nnz, w, out=out))
    print("K1: %.4f ms = %.0f GB/s = %.3f of 8 TB/s" % (ms, alg / ms / 1e6, alg / ms / 8e9), flush=True)
