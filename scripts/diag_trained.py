"""Diagnostic (not a test): what do FITTED weights look like to the exact top-k cascade?

Fits the WMRB model on planted-cluster, Zipf-popular interactions at the bench shape through the public API, and after
each block of epochs runs the exact top-10 over all users: step time, which stage 1 ran, refined pairs, kept superblocks
per user, flagged users -- plus the norm / bias distributions of the representations and a sample of them (fp16) for the
offline bound simulator (scripts/sim_cascade.py).  Output: gpurun_out/diag_trained.json (+ .npz)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import tensorrec_amd as T
from tensorrec_amd import ops
from tensorrec_amd.synth import planted_cluster_interactions

U = int(os.environ.get("U", 1_000_000)); I = int(os.environ.get("I", 1_000_000)); D = int(os.environ.get("D", 128))
NC = int(os.environ.get("NC", 256)); PER = int(os.environ.get("PER", 20))
BLOCKS = [int(x) for x in os.environ.get("EPOCH_BLOCKS", "5,15,30").split(",")]     # epochs added before each measurement
LR = float(os.environ.get("LR", 0.1)); S = int(os.environ.get("S", 100))
OUT = os.environ.get("OUT", "gpurun_out/diag_trained")
os.makedirs(os.path.dirname(OUT), exist_ok=True)

t0 = time.time()
inter, held, ucl, icl = planted_cluster_interactions(U, I, NC, PER, seed=0, holdout=0.05)
print("interactions %d (held out %d), most popular item %d, %.1f s" % (inter.nnz, held.nnz, int(np.bincount(inter.indices, minlength=I).max()), time.time() - t0), flush=True)
uf = sp.identity(U, dtype=np.float32, format="csr"); itf = sp.identity(I, dtype=np.float32, format="csr")
model = T.TensorRec(n_components=D, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
DEBUG = int(os.environ.get("DEBUG", 1))          # 0: no per-stage counters (they cost host syncs and extra passes): clean times
ops.FILTER_DEBUG = {} if DEBUG else None
res = {"shape": [U, I, D], "clusters": NC, "per_user": PER, "lr": LR, "n_sampled": S, "stages": []}


def q(x):
    x = x.float().reshape(-1)
    qs = torch.tensor([0.0, 0.01, 0.1, 0.5, 0.9, 0.99, 1.0], device=x.device)
    if x.numel() > 4_000_000:
        x = x[torch.randint(0, x.numel(), (4_000_000,), device=x.device)]
    return [round(float(v), 5) for v in torch.quantile(x, qs)]


def topk_run(label, **kw):
    if ops.FILTER_DEBUG is not None:
        ops.FILTER_DEBUG.clear()
    dts = []
    for rep in range(1 if ops.FILTER_DEBUG is not None else 2):      # the first call after a fit also pays for ~12 GB of hipMalloc
        torch.cuda.synchronize(); t = time.perf_counter()
        v, i = model.predict_top_k(uf, itf, k=10, user_batch_size=U, return_device=True, **kw)
        torch.cuda.synchronize(); dts.append(time.perf_counter() - t)
    dt = min(dts)
    return {"mode": label, "ms": 1e3 * dt, "ms_first_call": 1e3 * dts[0], "stats": dict(ops.LAST_FILTER_STATS), "debug": dict(ops.FILTER_DEBUG or {})}, v, i


total_epochs = 0
for blk in [0] + BLOCKS:
    if blk:
        t = time.perf_counter()
        model.fit_partial(inter, uf, itf, epochs=blk, learning_rate=LR, n_sampled_items=S)
        torch.cuda.synchronize()
        fit_s = time.perf_counter() - t
        total_epochs += blk
    else:
        model.fit_partial(inter, uf, itf, epochs=1, learning_rate=LR, n_sampled_items=S)     # builds the model; one step
        total_epochs += 1
        fit_s = None
    w = model.get_weights()
    wu, wi = torch.from_numpy(w["linear_weights_user_0"]).cuda(), torch.from_numpy(w["linear_weights_item"]).cuda()
    bu, bi = torch.from_numpy(w["user_feature_biases"]).cuda().reshape(-1), torch.from_numpy(w["item_feature_biases"]).cuda().reshape(-1)
    st = {"epochs": total_epochs, "fit_s": fit_s,
          "user_norm_q": q(wu.norm(dim=1)), "item_norm_q": q(wi.norm(dim=1)), "user_bias_q": q(bu), "item_bias_q": q(bi),
          "user_abs_elem_q": q(wu.abs()), "item_abs_elem_q": q(wi.abs())}
    # how clustered: cosine of item rows with their cluster mean (sampled)
    sel = torch.randint(0, I, (200_000,), device="cuda")
    cl = torch.from_numpy(icl).cuda()[sel]
    means = torch.zeros((NC, D), device="cuda").index_add_(0, cl, wi[sel])
    means = means / torch.bincount(cl, minlength=NC).clamp(min=1).unsqueeze(1)
    cosm = torch.nn.functional.cosine_similarity(wi[sel], means[cl], dim=1)
    st["item_cos_to_cluster_mean_q"] = q(cosm)
    st["item_resid_over_norm_q"] = q((wi[sel] - means[cl]).norm(dim=1) / wi[sel].norm(dim=1).clamp(min=1e-20))
    runs = []
    r, v1, i1 = topk_run("cascade (default)"); runs.append(r)
    T._native.set_tuning("topk_int8_prefilter", 0)
    r, v2, i2 = topk_run("bf16 filter only"); runs.append(r)
    T._native.set_tuning("topk_int8_prefilter", 1)
    st["cascade_equals_bf16_filter"] = bool(torch.equal(i1, i2) and torch.equal(v1, v2))
    st["runs"] = runs
    # held-out recall@10 on 20,000 users from the top-10 lists
    hu = np.unique(held.nonzero()[0])[:20000]
    top = i1[torch.from_numpy(hu).cuda()].cpu().numpy()
    hits = tot = 0
    for row, u in zip(top, hu):
        pos = held.indices[held.indptr[u]:held.indptr[u + 1]]
        hits += len(set(row.tolist()) & set(pos.tolist())); tot += len(pos)
    st["heldout_recall@10"] = hits / max(1, tot)
    res["stages"].append(st)
    print(json.dumps(st), flush=True)
    json.dump(res, open(OUT + ".json", "w"), indent=1)
    if total_epochs == 1 + sum(BLOCKS):
        rng = np.random.default_rng(1)
        us = np.sort(rng.choice(U, 2048, replace=False)); its = np.sort(rng.choice(I, min(I, 65536), replace=False))
        np.savez_compressed("%s_e%d.npz" % (OUT, total_epochs), users=w["linear_weights_user_0"][us].astype(np.float16),
                            items=w["linear_weights_item"][its].astype(np.float16),
                            user_bias=w["user_feature_biases"].reshape(-1)[us], item_bias=w["item_feature_biases"].reshape(-1)[its],
                            user_ids=us, item_ids=its, item_cluster=icl[its], user_cluster=ucl[us])
    del w, wu, wi
print("done %.1f s" % (time.time() - t0))
