"""Offline simulator of the exact top-k cascade's SELECTIVITY (CPU, torch; no GPU): for given user / item rows it evaluates
the int8 and bf16 stages' scores and bounds exactly as the kernels define them and reports what the capacities see --
refined (superblock, user) pairs, kept superblocks per user, survivors, flagged users -- under design options:

  layout    item order: none | norm | kmeans:C        (superblocks = 512 consecutive items of the order)
  residual  scores split as x.mu_s + x.(y - mu_s): both stages quantise the residual, the offset is fp32
  uscale    int8 user scale: global (min(4 rms, max)/127 over all users) | peruser
  eps16     bf16 bound: global item maxima (round 2) | per superblock

Not a test and not on any product path: a design tool (numbers quoted in DESIGN.md)."""
import os, sys, json, time, argparse
import numpy as np, torch

torch.set_num_threads(os.cpu_count() or 1)
SB = 512


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def kmeans_labels(y, C, iters=6, sample=32768, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = y.shape[0]
    idx = torch.randperm(n, generator=g)[:min(n, sample)]
    ys = y[idx]
    cent = ys[torch.randperm(ys.shape[0], generator=g)[:C]].clone()
    for _ in range(iters):
        a = (ys @ cent.t() - 0.5 * (cent * cent).sum(1)[None, :]).argmax(1)
        s = torch.zeros_like(cent).index_add_(0, a, ys)
        c = torch.bincount(a, minlength=C).float()
        cent = torch.where(c[:, None] > 0, s / c.clamp(min=1)[:, None], cent)
    lab = torch.empty(n, dtype=torch.int64)
    for s0 in range(0, n, 65536):
        lab[s0:s0 + 65536] = (y[s0:s0 + 65536] @ cent.t() - 0.5 * (cent * cent).sum(1)[None, :]).argmax(1)
    return lab, cent


def layout(y, how, bias=None):
    n = y.shape[0]
    if how == "none":
        return torch.arange(n)
    if how == "norm":
        return torch.argsort(y.norm(dim=1), descending=True)
    if how.startswith("kmeans:"):
        C = int(how.split(":")[1])
        lab, cent = kmeans_labels(y, C)
        # inside a cluster: by the projection on the cluster centre (norm-like, keeps near-duplicates together)
        proj = (y * cent[lab]).sum(1) / cent[lab].norm(dim=1).clamp(min=1e-20)
        key = lab.double() * 1e6 + (proj - proj.min()).double() / float(proj.max() - proj.min() + 1e-20) * 1e5
        return torch.argsort(key)
    if how.startswith("kmeans2:"):
        C1, C2 = [int(v) for v in how.split(":")[1:]]
        lab, cent = kmeans_labels(y, C1)
        key = torch.zeros(n, dtype=torch.float64)
        for c in range(C1):
            m = (lab == c).nonzero().reshape(-1)
            if m.numel() == 0:
                continue
            if m.numel() > 2 * SB:
                l2, _ = kmeans_labels(y[m] - cent[c], min(C2, max(2, m.numel() // SB)), seed=c)
            else:
                l2 = torch.zeros(m.numel(), dtype=torch.int64)
            key[m] = c * 1e4 + l2.double()
        return torch.argsort(key, stable=True)
    raise ValueError(how)


def simulate(X, Y, bu=None, bi=None, k=10, lay="none", residual=False, uscale="global", eps16="global", ksel=48, clip=4.0,
             verbose=False):
    nu, d = X.shape
    ni = Y.shape[0]
    ni = ni // SB * SB
    Y = Y[:ni]
    bu = torch.zeros(nu) if bu is None else bu.float()
    bi = torch.zeros(ni) if bi is None else bi[:ni].float()
    perm = layout(Y, lay)
    Y, bi = Y[perm].contiguous(), bi[perm].contiguous()
    n_sb = ni // SB
    K = d
    cK = (K + 6) * (2.0 ** -24 + 2.0 ** -22)
    S = X @ Y.t() + bu[:, None] + bi[None, :]                              # "truth" (fp32)
    tk = torch.topk(S, k, dim=1).values[:, -1]
    Ysb = Y.view(n_sb, SB, d)
    mu = Ysb.mean(1) if residual else torch.zeros(n_sb, d)
    R = (Ysb - mu[:, None, :]).reshape(ni, d)
    O = X @ mu.t()                                                        # [nu, n_sb] fp32 offsets
    xn = X.norm(dim=1)
    mun = mu.norm(dim=1)
    yn_sb = Y.norm(dim=1).view(n_sb, SB).max(1).values                      # full norms (reference chain's own rounding)
    # ---------------- int8 stage
    if uscale == "global":
        rms = float((X * X).mean().sqrt())
        a = torch.full((nu,), min(clip * rms, float(X.abs().max())) / 127.0)
    else:
        # per user: the better of no clipping and clipping at `clip` rms of the row (smaller measured error norm)
        a0 = X.abs().max(1).values / 127.0
        a1 = torch.minimum(a0, clip * (X * X).mean(1).sqrt() / 127.0)
        e0 = (X - (X / a0.clamp(min=1e-30)[:, None]).round().clamp(-127, 127) * a0[:, None]).norm(dim=1)
        e1 = (X - (X / a1.clamp(min=1e-30)[:, None]).round().clamp(-127, 127) * a1[:, None]).norm(dim=1)
        a = torch.where(e1 < e0, a1, a0)
    a = a.clamp(min=1e-30)
    qu = (X / a[:, None]).round().clamp(-127, 127)
    dxn = (X - qu * a[:, None]).norm(dim=1)
    bs = (R.abs().view(n_sb, -1).max(1).values / 127.0).clamp(min=1e-30)
    qi = (R / bs.repeat_interleave(SB)[:, None]).round().clamp(-127, 127)
    dR = R - qi * bs.repeat_interleave(SB)[:, None]
    rhat_n = (qi * bs.repeat_interleave(SB)[:, None]).norm(dim=1)
    Yh_s = (R.norm(dim=1) + dR.norm(dim=1)).view(n_sb, SB).max(1).values
    dY_s = dR.norm(dim=1).view(n_sb, SB).max(1).values
    I8 = (qu @ qi.t())                                                     # exact integers in fp32 (|acc| < 2^24)
    unit = a[:, None] * bs.repeat_interleave(SB)[None, :]
    bq = (bi[None, :] / unit).round().clamp(-2 ** 30, 2 ** 30)
    S8 = unit * (I8 + bq)
    dB = (bi[None, :] - unit * bq).abs().view(nu, n_sb, SB).max(2).values    # [nu, n_sb]
    M8 = S8.view(nu, n_sb, SB).max(2).values + O + bu[:, None]
    Bmax = float(bi.abs().max())
    e8 = 1.00195 * (xn[:, None] * (dY_s[None, :] + cK * yn_sb[None, :]) + dxn[:, None] * Yh_s[None, :] + dB
                    + cK * (bu.abs()[:, None] + Bmax) + cK * xn[:, None] * mun[None, :]) + 1e-30
    true_sbmax = S.view(nu, n_sb, SB).max(2).values
    viol8 = float(((M8 - true_sbmax).abs() - e8).max())
    LB8, UB8 = M8 - e8, M8 + e8
    tau8 = torch.topk(LB8, k, dim=1).values[:, -1]
    refined = UB8 >= tau8[:, None]
    out = {"layout": lay, "residual": residual, "uscale": uscale, "eps16": eps16, "users": nu, "items": ni,
           "int8_bound_violation": viol8, "refined_frac": float(refined.float().mean()),
           "refined_frac_max_over_superblocks": float(refined.float().mean(0).max()),
           "e8_over_topgap_median": float((e8.median(1).values / (tk - S.median(1).values).clamp(min=1e-20)).median())}
    # ---------------- bf16 stage on the refined pairs
    Xh, Rh = bf16(X), bf16(R)
    dx16 = (X - Xh).norm(dim=1)
    dr16 = (R - Rh).norm(dim=1)
    S16 = Xh @ Rh.t() + bi[None, :]
    S16 = S16.view(nu, n_sb, SB) + (O + bu[:, None])[:, :, None]
    M16 = S16.max(2).values
    rh_n = Rh.norm(dim=1)
    ck16 = (K + 2) * (2.0 ** -24 + 2.0 ** -22)
    if eps16 == "global":
        ni_, ai_ = float(rh_n.max()), float(dr16.max())
        e16 = (dx16 * ni_ * 1.0039 + xn * ai_ + ck16 * (xn * float(Y.norm(dim=1).max()) * 1.0078 + bu.abs() + Bmax)
               + ck16 * xn * float(mun.max())) * 1.00195 + 1e-30
        e16 = e16[:, None].expand(nu, n_sb)
    else:
        Rh_s = rh_n.view(n_sb, SB).max(1).values
        dR_s = dr16.view(n_sb, SB).max(1).values
        e16 = (dx16[:, None] * Rh_s[None, :] * 1.0039 + xn[:, None] * dR_s[None, :]
               + ck16 * (xn[:, None] * yn_sb[None, :] * 1.0078 + bu.abs()[:, None] + Bmax)
               + ck16 * xn[:, None] * mun[None, :]) * 1.00195 + 1e-30
    viol16 = float(((M16 - true_sbmax).abs() - e16).max())
    T = torch.where(refined, M16 - e16, LB8)                                 # the mixed table, in lower-bound space
    tau16 = torch.maximum(torch.topk(T, k, dim=1).values[:, -1], tau8)
    kept = refined & (M16 + e16 >= tau16[:, None])
    nkept = kept.sum(1)
    surv = (S16 + e16[:, :, None] >= tau16[:, None, None]) & kept[:, :, None]
    nsurv = surv.sum((1, 2))
    half = surv.view(nu, n_sb, 2, SB // 2).sum(3)
    list_over = (half > 8).any(2).any(1)
    flagged = (nkept > ksel) | (nsurv > 64) | list_over
    # sanity: the true top-k must be among the survivors
    topi = torch.topk(S, k, dim=1).indices
    lost = int((~surv.view(nu, ni).gather(1, topi)).sum())
    out.update({"bf16_bound_violation": viol16, "kept_mean": float(nkept.float().mean()),
                "kept_q50_90_99_max": [float(v) for v in torch.quantile(nkept.float(), torch.tensor([0.5, 0.9, 0.99, 1.0]))],
                "survivors_mean": float(nsurv.float().mean()),
                "survivors_q50_90_99_max": [float(v) for v in torch.quantile(nsurv.float(), torch.tensor([0.5, 0.9, 0.99, 1.0]))],
                "flagged_frac": float(flagged.float().mean()), "flag_kept": float((nkept > ksel).float().mean()),
                "flag_surv": float((nsurv > 64).float().mean()), "flag_list": float(list_over.float().mean()),
                "topk_items_lost": lost,
                "resid_norm_over_norm_median": float((R.norm(dim=1) / Y.norm(dim=1).clamp(min=1e-20)).median())})
    return out


def make(kind, n, d, g):
    x = torch.randn((n, d), generator=g)
    if kind == "normalised": x = x / x.norm(dim=1, keepdim=True)
    elif kind == "heavy_tail": x = x * torch.exp(1.5 * torch.randn((n, d), generator=g))
    elif kind == "sparse": x = x * (torch.rand((n, d), generator=g) < 0.1)
    elif kind == "integers": x = torch.round(x * 2)
    elif kind == "clustered": x = torch.randn((8, d), generator=g)[torch.randint(0, 8, (n,), generator=g)] + 0.05 * x
    elif kind.startswith("clustered256_"):
        x = torch.randn((256, d), generator=g)[torch.randint(0, 256, (n,), generator=g)] + float(kind.split("_")[1]) / 10.0 * x
    elif kind == "scaled_rows": x = x * torch.exp(2.0 * torch.randn((n, 1), generator=g))
    return x.contiguous()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="gauss,normalised,heavy_tail,sparse,integers,clustered,clustered256_03,clustered256_10,scaled_rows")
    ap.add_argument("--npz", default=None, help="sampled fitted rows from scripts/diag_trained.py")
    ap.add_argument("--users", type=int, default=512)
    ap.add_argument("--items", type=int, default=131072)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--configs", default="none/0/global/global,kmeans:256/0/peruser/sb,kmeans:256/1/peruser/sb")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(1)
    cases = []
    if args.npz:
        z = np.load(args.npz)
        cases.append(("fitted:" + os.path.basename(args.npz), torch.from_numpy(z["users"].astype(np.float32))[:args.users],
                      torch.from_numpy(z["items"].astype(np.float32))[:args.items], torch.from_numpy(z["user_bias"])[:args.users],
                      torch.from_numpy(z["item_bias"])[:args.items]))
    else:
        for kind in args.kinds.split(","):
            cases.append((kind, make(kind, args.users, args.d, g), make(kind, args.items, args.d, g), None, None))
    res = []
    for name, X, Y, bu, bi in cases:
        for cfg in args.configs.split(","):
            lay, resid, us, e16 = cfg.split("/")
            t = time.time()
            r = simulate(X, Y, bu, bi, lay=lay, residual=resid == "1", uscale=us, eps16=e16)
            r["kind"] = name; r["seconds"] = time.time() - t
            print(json.dumps(r), flush=True)
            res.append(r)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
