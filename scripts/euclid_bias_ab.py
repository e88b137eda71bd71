"""Users the Euclidean top-k route has to re-do on the fp32 path, with the item biases inside the cascade's ordering (h = u.i - r_i / 2 +
lambda b_i, the default) and without (euclid_bias_in_order = 0): 700 users x 280,000 items, d = 128, k = 10, item / user biases of
several sizes (distances are ~16)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops, _native as N

rng = np.random.default_rng(0)
n_u, n_i, d, k = 700, 280_000, 128, 10
u = torch.from_numpy(rng.standard_normal((n_u, d)).astype(np.float32)).cuda()
v = torch.from_numpy((rng.standard_normal((n_i, d)) * rng.uniform(0.7, 1.3, (n_i, 1))).astype(np.float32)).cuda()
out = {}
for scale in (0.001, 0.02, 0.2, 2.0):
    ub = torch.from_numpy((scale * rng.standard_normal(n_u)).astype(np.float32)).cuda()
    ib = torch.from_numpy((scale * rng.standard_normal(n_i)).astype(np.float32)).cuda()
    row = {}
    res = {}
    for knob in (0, 1):
        N.set_tuning("euclid_bias_in_order", knob)
        res[knob] = ops.score_topk_euclid_filtered(u, v, k, ub, ib)
        row["in_order=%d" % knob] = int(ops.LAST_FILTER_STATS["euclid_uncertified_users"])
    row["identical"] = bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
    for pct in [int(x) for x in os.environ.get("PCTS", "").split(",") if x]:
        N.set_tuning("euclid_lambda_pct", pct)
        r = ops.score_topk_euclid_filtered(u, v, k, ub, ib)
        row["lambda_pct=%d" % pct] = int(ops.LAST_FILTER_STATS["euclid_uncertified_users"])
        row["identical"] = row["identical"] and bool(torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1]))
    N.set_tuning("euclid_lambda_pct", ops.EUCLID_LAMBDA_PCT)
    out["bias_sigma=%g" % scale] = row
N.set_tuning("euclid_bias_in_order", 1)
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/euclid_bias_ab.json", "w"), indent=1)
