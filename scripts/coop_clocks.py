"""Diagnostics: where the single-kernel step of configs[1] spends its time (wall_clock64 stamps of workgroup 0 at the phase boundaries;
100 MHz).  python scripts/coop_clocks.py [wg_per_cu]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import bench_records as BR
import tensorrec_amd as T
import tensorrec_amd.tensorrec as TT
if len(sys.argv) > 1:
    T._native.set_tuning("coop_wg_per_cu", int(sys.argv[1]))
T._native.set_tuning("coop_clocks", 1)
rng = np.random.default_rng(0)
n_users, n_items, d, S, lr = 943, 1682, 64, 168, 0.05
inter = BR._zipf_interactions(n_users, n_items, 160, rng, exponent=1.0)
uf = sp.identity(n_users, dtype=np.float32, format="csr")
itf = BR._side_features(n_items, 19, 3, rng)
model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0)
holder = {}
orig = TT._CoopStep.run
def run(self, *a, **k):
    holder["ws"] = self.ws
    return orig(self, *a, **k)
TT._CoopStep.run = run
model.fit_partial(inter, uf, itf, epochs=20, learning_rate=lr, n_sampled_items=S)
torch.cuda.synchronize()
ws = holder["ws"]
off = (ws.numel() - 64 + 1) & ~1
clk = ws[off:off + 16].view(torch.int64).cpu().numpy()
names = ["phase 1 (item tower fwd, clears)", "barrier", "phase 2 (users)", "barrier", "phase 3 (G^T U)", "barrier", "phase 4 (item tower bwd + Adam)"]
print("step form", model.last_step_form)
for i, n in enumerate(names):
    print("%-36s %7.2f us" % (n, (clk[i + 1] - clk[i]) / 100.0))
print("%-36s %7.2f us" % ("total (workgroup 0)", (clk[7] - clk[0]) / 100.0))
