"""Why does the int8 stage (blockmax_i8x16_kernel, power-capped) take 76 ms on the headline's Gaussian rows and 86 ms on
fitted rows?  The SAME kernel, launched back to back for a few seconds on each operand set at 1M x 1M x 128, with rocm-smi's
shader clock and socket power sampled beside it.  Operand sets:

  gaussian          the headline's L2-normalised N(0,1) rows (through the product's own preparation)
  fitted            20 WMRB epochs on planted-cluster Zipf interactions (bench_records.trained_weights_record's weights)
  fitted_shuffled   the fitted int8 rows, item rows in a random order (same bytes, no order structure in the catalogue)
  fitted_elem_perm  the fitted int8 rows with the k-columns of every row permuted independently (same value histogram per row,
                    no direction structure -> if toggling of accumulators matters this changes, magnitudes do not)
  zero              all-zero int8 operands (no toggling at all)
  pm127             every int8 element +-127 at random (the most toggling int8 allows)
  gaussian_half     the Gaussian int8 rows arithmetically halved (|q| <= 63: one bit of range traded for magnitude)

Per set: ms per launch (HIP events), median sclk, median socket power, rows of the user layout.  -> gpurun_out/i8_data_dependence.json
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

from tensorrec_amd import ops
from tensorrec_amd import _native as N

U = int(os.environ.get("U", 1_000_000))
I = int(os.environ.get("I", 1_000_000))
d, k = 128, 10
SECONDS = float(os.environ.get("SECONDS", 5.0))
dev = torch.device("cuda", 0)


class Smi(threading.Thread):
    """rocm-smi sclk / socket power every ~0.3 s while alive."""
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True,
                                     timeout=5).stdout
                clk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
                if clk and pw:
                    self.samples.append((time.perf_counter(), int(clk.group(1)), float(pw.group(1))))
            except Exception:
                pass
            time.sleep(0.3)


def prepare(w_u, w_i, beta_u, beta_i):
    """The product's own operand preparation (the bench step's first half) -> (uop, iop, user bias in layout order, item bias)."""
    ub = beta_u.reshape(-1).contiguous()
    ib = beta_i.reshape(-1).contiguous()
    uop = ops.score_prep_filter(w_u, sort_users=True, k=k, user_bias=ub)
    iop = ops.score_prep_filter(w_i, bias=ib, want_gstats=True)
    ops.score_prep_i8_pair(uop, iop, ib, ops.SUPERBLOCK_ROWS, 10)
    return uop, iop, (uop.bias_sorted if uop.bias_sorted is not None else ub), ib


def run_i8(uop, iop, user_bias, seconds):
    """Exactly the launch of ops._cascade_stage1, in a loop."""
    sb_rows = ops.SUPERBLOCK_ROWS
    n_u, n_i, kpad = uop.n, iop.n, uop.kpad
    n_sb = (n_i + sb_rows - 1) // sb_rows
    rows_wg = N.query("trec_score_rows_per_workgroup", ops.DTYPE_BF16, kpad)
    rblocks = (n_u + rows_wg - 1) // rows_wg
    n_chunks = max(1, min(n_sb, -(-32 * 768 // rblocks)))
    user_err = torch.empty((n_u, 4), dtype=torch.float32, device=dev)
    N.call("trec_score_user_err_i8", N.ptr(uop.stats8), N.ptr(user_bias), N.ptr(iop.gstats8), kpad, n_u, N.ptr(iop.scales),
           N.ptr(uop.wg_scale), int(uop.wg_rows or 0), N.ptr(user_err))
    stride = (n_u + 3) // 4 * 4
    _, n_ch = ops.blockmax_i8_chunks(n_i, n_chunks, sb_rows)
    table = torch.empty((n_sb, stride), dtype=torch.float32, device=dev)
    chunk_top = torch.empty((n_ch * 10, stride), dtype=torch.float32, device=dev)

    def launch():
        N.call("trec_score_gemm_blockmax_i8", N.ptr(uop.i8), N.ptr(iop.i8), kpad, n_u, n_i, N.ptr(user_bias),
               N.ptr(iop.bias_q), N.ptr(iop.scales), N.ptr(iop.sb_stats), sb_rows, n_chunks, N.ptr(table), stride,
               N.ptr(user_err), N.ptr(chunk_top), 10, N.ptr(uop.wg_scale), N.ptr(uop.wg_class), int(uop.wg_rows or 0))
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    smi = Smi()
    smi.start()
    t_start = time.perf_counter()
    times = []
    while time.perf_counter() - t_start < seconds:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        evs[0].record()
        for j in range(8):
            launch()
            evs[j + 1].record()
        torch.cuda.synchronize()
        times += [evs[j].elapsed_time(evs[j + 1]) for j in range(8)]
    t_end = time.perf_counter()
    smi.stop = True
    smi.join()
    inside = [(c, p) for t, c, p in smi.samples if t_start + 1.0 <= t <= t_end]
    return {"ms_per_launch_mean": float(np.mean(times)), "ms_per_launch_min": float(np.min(times)),
            "ms_first_8": [round(x, 2) for x in times[:8]], "ms_last_8": [round(x, 2) for x in times[-8:]], "launches": len(times),
            "sclk_mhz_median": float(np.median([c for c, _ in inside])) if inside else None,
            "socket_power_w_median": float(np.median([p for _, p in inside])) if inside else None,
            "smi_samples": len(inside), "user_layout_rows": int(n_u),
            "mean_abs_q_users": float(uop.i8[:: 97].float().abs().mean().item()),
            "mean_abs_q_items": float(iop.i8[:: 97].float().abs().mean().item())}


def fitted_weights():
    import tensorrec_amd as T
    from tensorrec_amd.synth import planted_cluster_interactions
    inter, _, _, _ = planted_cluster_interactions(U, I, 256, 20, seed=0, holdout=0.05, device=dev)
    uf = sp.identity(U, dtype=np.float32, format="csr")
    itf = sp.identity(I, dtype=np.float32, format="csr")
    model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
    model.fit_partial(inter, uf, itf, epochs=20, learning_rate=0.1, n_sampled_items=100)
    w = model.get_weights()
    del model
    torch.cuda.empty_cache()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(w["linear_weights_user_0"]), t(w["linear_weights_item"]), t(w["user_feature_biases"].reshape(-1, 1)),
            t(w["item_feature_biases"].reshape(-1, 1)))


def main():
    out = {"shape": [U, I, d], "seconds_per_set": SECONDS, "sets": {}}
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    w_u = ops.l2_normalize_rows(torch.randn((U, d), device=dev, generator=gen))
    gi = torch.Generator(device=dev)
    gi.manual_seed(1)
    w_i = ops.l2_normalize_rows(torch.randn((I, d), device=dev, generator=gi))
    beta_u = 0.05 * torch.randn((U, 1), device=dev, generator=gen)
    beta_i = 0.05 * torch.randn((I, 1), device=dev, generator=gi)

    uop, iop, ub, ib = prepare(w_u, w_i, beta_u, beta_i)
    out["sets"]["gaussian"] = run_i8(uop, iop, ub, SECONDS)
    print("gaussian", out["sets"]["gaussian"], flush=True)
    keep_u, keep_i = uop.i8.clone(), iop.i8.clone()
    uop.i8.copy_(torch.div(keep_u, 2, rounding_mode="trunc"))
    iop.i8.copy_(torch.div(keep_i, 2, rounding_mode="trunc"))
    out["sets"]["gaussian_half"] = run_i8(uop, iop, ub, SECONDS)
    print("gaussian_half", out["sets"]["gaussian_half"], flush=True)
    uop.i8.zero_()
    iop.i8.zero_()
    out["sets"]["zero"] = run_i8(uop, iop, ub, SECONDS)
    print("zero", out["sets"]["zero"], flush=True)
    uop.i8.copy_((torch.randint(0, 2, uop.i8.shape, device=dev, generator=gen, dtype=torch.int8) * 2 - 1) * 127)
    iop.i8.copy_((torch.randint(0, 2, iop.i8.shape, device=dev, generator=gen, dtype=torch.int8) * 2 - 1) * 127)
    out["sets"]["pm127"] = run_i8(uop, iop, ub, SECONDS)
    print("pm127", out["sets"]["pm127"], flush=True)
    uop.i8.copy_(keep_u)
    iop.i8.copy_(keep_i)
    out["sets"]["gaussian_again"] = run_i8(uop, iop, ub, SECONDS)           # (thermal / order control: same as the first set)
    print("gaussian_again", out["sets"]["gaussian_again"], flush=True)
    del uop, iop, keep_u, keep_i, w_u, w_i
    torch.cuda.empty_cache()

    fw_u, fw_i, fb_u, fb_i = fitted_weights()
    uop, iop, ub, ib = prepare(fw_u, fw_i, fb_u, fb_i)
    out["sets"]["fitted"] = run_i8(uop, iop, ub, SECONDS)
    print("fitted", out["sets"]["fitted"], flush=True)
    keep_i = iop.i8.clone()
    perm = torch.randperm(I, device=dev, generator=gen)
    iop.i8.copy_(keep_i[perm])
    out["sets"]["fitted_shuffled"] = run_i8(uop, iop, ub, SECONDS)
    print("fitted_shuffled", out["sets"]["fitted_shuffled"], flush=True)
    cols = torch.argsort(torch.rand((I, iop.i8.shape[1]), device=dev, generator=gen), dim=1)
    iop.i8.copy_(torch.gather(keep_i, 1, cols))
    del cols
    out["sets"]["fitted_elem_perm"] = run_i8(uop, iop, ub, SECONDS)
    print("fitted_elem_perm", out["sets"]["fitted_elem_perm"], flush=True)
    # the same fitted rows in the CALLER's order with one scale for all (no class sort): layout without class padding
    iop.i8.copy_(keep_i)
    out["sets"]["fitted_again"] = run_i8(uop, iop, ub, SECONDS)
    print("fitted_again", out["sets"]["fitted_again"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/i8_data_dependence.json", "w"), indent=1)
    print(json.dumps({n_: (v["ms_per_launch_mean"], v["sclk_mhz_median"], v["socket_power_w_median"]) for n_, v in out["sets"].items()}))


if __name__ == "__main__":
    main()
