"""The int8 stage-0 kernel alone, N launches at the bench shape (power / clock traces, PMC passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorrec_amd import ops, _native as N
U = I = 1_000_000; d = 128
n = int(os.environ.get("LOOPS", 3)); which = os.environ.get("KERNEL", "i8"); TK = int(os.environ.get("TK", 10))
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
uop = ops.score_prep_filter(u); iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
ops.score_prep_i8_pair(uop, iop, ib)
N.load().trec_set_tuning(b"blockmax_i8_mfma", int(os.environ.get("MFMA", 1)))
N.load().trec_set_tuning(b"blockmax_i8_users", int(os.environ.get("USERS", 192)))
N.load().trec_set_tuning(b"blockmax_i8_waves", int(os.environ.get("WAVES", 4)))
N.load().trec_set_tuning(b"blockmax_i8_tile", int(os.environ.get("TILE", 128)))
n_sb = (I + 511) // 512
table = torch.empty((n_sb, U), dtype=torch.float32, device="cuda")
uerr = torch.empty((U, 3), device="cuda")
N.call("trec_score_user_err_i8", N.ptr(uop.stats8), N.ptr(ub), N.ptr(iop.gstats8), d, U, N.ptr(uerr))
ctop = torch.empty((13 * 10, U), device="cuda")
def run():
    if which == "i8":
        N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), d, U, I, N.ptr(ub), N.ptr(iop.bias_q),
               N.ptr(iop.scales), N.ptr(iop.sb_stats), 512, 13, N.ptr(table), U, N.ptr(uerr) if TK else None,
               N.ptr(ctop) if TK else None, TK)
    else:
        N.call("trec_score_gemm_blockmax", N.ptr(uop.bf16), N.ptr(iop.bf16), ops.DTYPE_BF16, d, U, I, N.ptr(ub), N.ptr(ib),
               ops.MODE_DOT, None, None, 512, 13, N.ptr(table), U, int(os.environ.get("VARIANT", 1)))
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): run()
torch.cuda.synchronize()
print("%s (TK=%d, MFMA=%s): %.2f ms per launch over %d launches" % (which, TK, os.environ.get("MFMA", "1") + " VARIANT=" + os.environ.get("VARIANT", "1") + " USERS=" + os.environ.get("USERS", "192") + " WAVES=" + os.environ.get("WAVES", "4") + " TILE=" + os.environ.get("TILE", "128"), (time.perf_counter() - t0) / n * 1e3, n))
