"""How much of the int8 stage's time is the DATA?  The stage sits at the chip's power cap (77 ms on Gaussian rows, 63 ms
on zeros at 1M x 1M): this probe runs the stage's own launch (trec_score_gemm_blockmax_i8 behind ops.score_prep_i8_pair) on operands
whose QUANTISED values follow different distributions -- full-range Gaussian, narrow (|q| <= 31 next to one full-scale component per
row), non-negative items, zeros -- and prints the average launch time of 10 back-to-back launches each."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrec_amd import ops, ops_topk as OT, _native as N

U = int(os.environ.get("U", 1_000_000)); I = int(os.environ.get("I", 1_000_000)); d = 128
g = torch.Generator(device="cuda"); g.manual_seed(0)

def ints(n, lo, hi):
    return torch.randint(lo, hi + 1, (n, d), device="cuda", generator=g).float()

def with_outlier(x, v=127.0):
    x = x.clone(); x[:, 0] = v; return x

kinds = {
    "gaussian": (torch.randn((U, d), device="cuda", generator=g), torch.randn((I, d), device="cuda", generator=g)),
    "uniform +-127": (ints(U, -127, 127), ints(I, -127, 127)),
    "narrow +-31 (+ one full-scale component)": (with_outlier(ints(U, -31, 31)), with_outlier(ints(I, -31, 31))),
    "narrow +-7 (+ one full-scale component)": (with_outlier(ints(U, -7, 7)), with_outlier(ints(I, -7, 7))),
    "items non-negative 0..127": (ints(U, -127, 127), ints(I, 0, 127)),
    "both non-negative 0..127": (ints(U, 0, 127), ints(I, 0, 127)),
    "zeros (+ one full-scale component)": (with_outlier(torch.zeros((U, d), device="cuda")), with_outlier(torch.zeros((I, d), device="cuda"))),
}
out = {"users": U, "items": I, "d": d}
sb_rows, k = OT.SUPERBLOCK_ROWS, 10
for name, (u, v) in kinds.items():
    uop = ops.score_prep_filter(u, sort_users=True, k=k)
    iop = ops.score_prep_filter(v, want_gstats=True)
    OT.score_prep_i8_pair(uop, iop, None, sb_rows, 10)
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad
    n_sb = (n_i + sb_rows - 1) // sb_rows
    rows_wg = N.query("trec_score_rows_per_workgroup", ops.DTYPE_BF16, kpad)
    rblocks = (n_u + rows_wg - 1) // rows_wg
    n_chunks = max(2, min(n_sb, -(-32 * 768 // rblocks)))
    user_err = torch.empty((n_u, 4), dtype=torch.float32, device="cuda")
    N.call("trec_score_user_err_i8", N.ptr(uop.stats8), None, N.ptr(iop.gstats8), kpad, n_u, N.ptr(iop.scales),
           N.ptr(uop.wg_scale), int(uop.wg_rows or 0) if uop.wg_scale is not None else 0, N.ptr(user_err))
    stride = (n_u + 3) // 4 * 4
    chunk_len, n_ch = OT.blockmax_i8_chunks(n_i, n_chunks, sb_rows)
    table = torch.empty((n_sb, stride), dtype=torch.float32, device="cuda")
    chunk_top = torch.empty((n_ch * 10, stride), dtype=torch.float32, device="cuda")
    def launch():
        N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), kpad, n_u, n_i, None, N.ptr(iop.bias_q), N.ptr(iop.scales),
               N.ptr(iop.sb_stats), sb_rows, n_chunks, N.ptr(table), stride, N.ptr(user_err), N.ptr(chunk_top), 10,
               N.ptr(uop.wg_scale), N.ptr(uop.wg_class), int(uop.wg_rows or 0) if uop.wg_scale is not None else 0)
    for _ in range(3): launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for j in range(10):
        launch(); ev[j + 1].record()
    torch.cuda.synchronize()
    ms = [ev[j].elapsed_time(ev[j + 1]) for j in range(10)]
    q = uop.i8.view(torch.int8).float()
    out[name] = {"avg_launch_ms": sum(ms) / 10, "min": min(ms), "max": max(ms), "user_q_abs_mean": float(q.abs().mean().item())}
    print(name, out[name], flush=True)
    del uop, iop, table, chunk_top, u, v
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/int8_power_probe.json", "w"), indent=1)
