"""Offline (CPU) answer to "would a narrower first stage -- MXFP8 / MXFP6 / MXFP4 on v_mfma_scale_f32_*_f8f6f4, twice the int8
MFMA rate for FP6 / FP4 -- pay in the exact top-k cascade?" (VERDICT r5 next #9).

For given user / item rows it quantises both operands to each format exactly as the hardware would take them (OCP MX: blocks of 32
along K share a power-of-two scale, elements round-to-nearest-even and saturate; int8: one scale per user row and per item
superblock, the shipped stage), contracts them in fp64, and applies the SAME proven bound the int8 stage ships with
(|x.y - xq.yq| <= |x| |dy| + |dx| |yq|, item-side norms as superblock maxima, csrc/topk_cascade.hip) to get, per superblock size,

  kept   the share of (superblock, user) pairs the stage cannot rule out = the share of ALL pairs the bf16 stage must re-score,
  table  the bytes of the [n_sb, users] table of superblock maxima at 1M x 1M,

and a step-time model next to them: stage 1 at the format's MFMA rate (int8 measured; FP8 same rate; FP6 / FP4 twice) + the bf16
refinement at its measured cost per refined pair.  Not a test and not on any product path."""
import argparse
import json
import os
import time

import numpy as np
import torch

torch.set_num_threads(os.cpu_count() or 1)

# OCP microscaling element formats: (exponent bits, mantissa bits, largest normal, emax of the element)
MX = {"fp8_e4m3": (4, 3, 448.0, 8), "fp6_e2m3": (2, 3, 7.5, 2), "fp6_e3m2": (3, 2, 28.0, 4), "fp4_e2m1": (2, 1, 6.0, 2)}
INT8_STAGE_MS = 79.3          # measured, 1M x 1M x 128 (profiles/r05_kernel_stats.csv)
RATE_VS_INT8 = {"int8": 1.0, "fp8_e4m3": 1.0, "fp6_e2m3": 2.0, "fp6_e3m2": 2.0, "fp4_e2m1": 2.0}
REFINE_MS_PER_PERCENT = 3.9   # measured: 5.5 ms of bf16 refinement + compaction for 1.43 % of the pairs (profiles/r05_bench_full.json)


def grid(ebits, mbits, vmax):
    """all non-negative values of a (sign, ebits, mbits) minifloat without inf / nan codes, up to vmax"""
    bias = (1 << (ebits - 1)) - 1
    vals = {0.0}
    for e in range(0, 1 << ebits):
        for m in range(0, 1 << mbits):
            v = (m / (1 << mbits)) * 2.0 ** (1 - bias) if e == 0 else (1 + m / (1 << mbits)) * 2.0 ** (e - bias)
            if v <= vmax:
                vals.add(v)
    return torch.tensor(sorted(vals), dtype=torch.float64)


def mx_quantise(x, fmt, block=32):
    """x [n, d] -> dequantised MX image (float64): per block of `block` along d, scale 2^(floor(log2 max|x|) - emax), elements RNE"""
    ebits, mbits, vmax, emax = MX[fmt]
    g = grid(ebits, mbits, vmax)
    n, d = x.shape
    xb = x.double().reshape(n, d // block, block)
    amax = xb.abs().amax(dim=2, keepdim=True).clamp(min=1e-300)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = (xb / scale).clamp(-vmax, vmax)
    # nearest grid value (ties: to the even mantissa = the lower index of equal distances alternates; measure-zero here)
    idx = torch.bucketize(y.abs(), g)
    lo = g[(idx - 1).clamp(min=0)]
    hi = g[idx.clamp(max=g.numel() - 1)]
    q = torch.where((y.abs() - lo) <= (hi - y.abs()), lo, hi) * torch.sign(y)
    return (q * scale).reshape(n, d)


def int8_users(x):
    a = (x.abs().amax(dim=1, keepdim=True) / 127.0).clamp(min=1e-30)
    return ((x / a).round().clamp(-127, 127) * a).double()


def int8_items(y, sb):
    n, d = y.shape
    ys = y.reshape(n // sb, sb * d)
    b = (ys.abs().amax(dim=1, keepdim=True) / 127.0).clamp(min=1e-30)
    return ((ys / b).round().clamp(-127, 127) * b).reshape(n, d).double()


def stage(X, Y, fmt, sb, k, S_true):
    nu, d = X.shape
    ni = Y.shape[0] // sb * sb
    Y = Y[:ni]
    if fmt == "int8":
        Xq, Yq = int8_users(X), int8_items(Y, sb)
    else:
        Xq, Yq = mx_quantise(X, fmt), mx_quantise(Y, fmt)
    n_sb = ni // sb
    Sq = (Xq @ Yq.t()).float()
    M = Sq.view(nu, n_sb, sb).amax(dim=2)
    xn = X.double().norm(dim=1)
    dxn = (X.double() - Xq).norm(dim=1)
    dY_s = (Y.double() - Yq).norm(dim=1).view(n_sb, sb).amax(dim=1)
    Yq_s = Yq.norm(dim=1).view(n_sb, sb).amax(dim=1)
    cK = (d + 6) * (2.0 ** -24 + 2.0 ** -22)
    e = (1.002 * (xn[:, None] * dY_s[None, :] + dxn[:, None] * Yq_s[None, :] + cK * xn[:, None] * Yq_s[None, :])).float()
    true_max = S_true[:, :ni].view(nu, n_sb, sb).amax(dim=2)
    viol = float(((M - true_max).abs() - e).max())
    tau = torch.topk(M - e, k, dim=1).values[:, -1]
    kept = (M + e) >= tau[:, None]
    return {"format": fmt, "superblock": sb, "kept": float(kept.float().mean()), "bound_violation": viol,
            "e_over_gap_median": float((e.median(dim=1).values / (tau - M.median(dim=1).values).clamp(min=1e-20)).median()),
            "table_GB_at_1Mx1M": 1e6 * (1e6 / sb) * 4 / 1e9}


def model_ms(r):
    s1 = INT8_STAGE_MS / RATE_VS_INT8[r["format"]]
    return s1 + REFINE_MS_PER_PERCENT * 100.0 * r["kept"]


def make(kind, n, d, g):
    x = torch.randn((n, d), generator=g)
    if kind == "normalised":
        x = x / x.norm(dim=1, keepdim=True)
    elif kind.startswith("clustered256_"):
        x = torch.randn((256, d), generator=g)[torch.randint(0, 256, (n,), generator=g)] + float(kind.split("_")[1]) / 10.0 * x
    return x.contiguous()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="normalised,clustered256_03")
    ap.add_argument("--npz", default=None, help="sampled fitted rows (scripts/diag_trained.py)")
    ap.add_argument("--users", type=int, default=512)
    ap.add_argument("--items", type=int, default=131072)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--superblocks", default="512,128,32")
    ap.add_argument("--formats", default="int8,fp8_e4m3,fp6_e2m3,fp6_e3m2,fp4_e2m1")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(1)
    cases = []
    if args.npz:
        z = np.load(args.npz)
        cases.append(("fitted:" + os.path.basename(args.npz), torch.from_numpy(z["users"].astype(np.float32))[:args.users],
                      torch.from_numpy(z["items"].astype(np.float32))[:args.items]))
    for kind in [k_ for k_ in args.kinds.split(",") if k_]:
        cases.append((kind, make(kind, args.users, args.d, g), make(kind, args.items, args.d, g)))
    res = []
    for name, X, Y in cases:
        S_true = (X.double() @ Y.double().t()).float()
        for sb in [int(v) for v in args.superblocks.split(",")]:
            for fmt in args.formats.split(","):
                t = time.time()
                r = stage(X, Y, fmt, sb, args.k, S_true)
                r.update({"rows": name, "users": X.shape[0], "items": Y.shape[0], "model_step_ms": model_ms(r),
                          "seconds": time.time() - t})
                print(json.dumps(r), flush=True)
                res.append(r)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
