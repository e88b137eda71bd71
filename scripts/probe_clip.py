"""User clip level of the int8 quantisation vs refined pairs / step time (per-superblock item scales)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorrec_amd import ops, _native as N
U = I = 1_000_000; d = 128; k = 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = ops.l2_normalize_rows(torch.randn((U, d), device="cuda", generator=g))
v = ops.l2_normalize_rows(torch.randn((I, d), device="cuda", generator=g))
ub = torch.zeros(U, device="cuda"); ib = torch.zeros(I, device="cuda")
out = {}
for clip in (40, 35, 30, 45, 40):
    N.load().trec_set_tuning(b"i8_user_clip_x10", clip)
    def step():
        uop = ops.score_prep_filter(u); iop = ops.score_prep_filter(v, bias=ib, want_gstats=True)
        return ops.score_topk_filtered(uop, iop, k, ub, ib, prefilter="int8")
    step(); torch.cuda.synchronize()
    ops.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    i8 = sum(a.elapsed_time(b) for n, a, b in ev if n == "score_gemm_blockmax_i8") / 3
    gr = sum(a.elapsed_time(b) for n, a, b in ev if n == "score_gemm_blockmax_grouped") / 3
    print("clip %.1f rms: step %.2f ms, int8 %.2f, grouped %.2f, refined rows %d, flagged %d" % (clip / 10, ms, i8, gr, ops.LAST_FILTER_STATS["refined_rows"], ops.LAST_FILTER_STATS["flagged_users"]))
