"""K1 on 1M identity rows, d = 128: the plain gather, the gather with the filter-operand epilogue, and gather + separate prep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from tensorrec_amd import ops
from tensorrec_amd.sparse import SparseFeatures
U, d = 1_000_000, 128
f = SparseFeatures(sp.identity(U, dtype=np.float32, format="csr"), "cuda")
w = torch.randn((U, d), device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
plain = t(lambda: ops.spmm_raw(f.indptr, f.indices, f.values, None, U, f.nnz, w))
one = t(lambda: ops.spmm_raw(f.indptr, f.indices, f.values, None, U, f.nnz, w, one_per_row=True))
print("one-per-row kernel %.4f ms %.0f GB/s" % (one, (U * 8 + 2 * U * d * 4) / one / 1e6))
fused = t(lambda: ops.spmm_filter_operand(f, w, want_gstats=True))
r = ops.spmm_raw(f.indptr, f.indices, f.values, None, U, f.nnz, w)
prep = t(lambda: ops.score_prep_filter(r, want_gstats=True))
b_plain = U * 8 + (U + 1) * 8 + 2 * U * d * 4
b_fused = b_plain + U * d * 2 + U * 8
print("plain %.4f ms %.0f GB/s | fused %.4f ms %.0f GB/s | separate prep %.4f ms (plain + prep %.4f)" % (
    plain, b_plain / plain / 1e6, fused, b_fused / fused / 1e6, prep, plain + prep))
