"""Ranks of the positives of a [512, 1M] score slab (20 per user): wave-per-pair counting vs one pass per user."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tensorrec_amd import ops


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


nu, ni, per = 512, 1_000_000, 20
rng = np.random.default_rng(0)
slab = torch.randn((nu, ni), device="cuda")
xu = torch.arange(nu, device="cuda", dtype=torch.int32).repeat_interleave(per)
xi = torch.from_numpy(np.sort(rng.integers(0, ni, (nu, per)), axis=1).reshape(-1).astype(np.int32)).cuda()
indptr = torch.arange(0, (nu + 1) * per, per, device="cuda", dtype=torch.int64)
tgt = slab[xu.long(), xi.long()].contiguous()
a = ops.rank_of_pairs(slab, 0, 0, ni, xu, xi, tgt)
b = ops.rank_of_pairs_by_user(slab, 0, 0, ni, indptr, xi, tgt)
assert torch.equal(a, b)
t1 = timeit(lambda: ops.rank_of_pairs(slab, 0, 0, ni, xu, xi, tgt))
t2 = timeit(lambda: ops.rank_of_pairs_by_user(slab, 0, 0, ni, indptr, xi, tgt))
print("512 users x 1M items, 20 positives each: wave per pair %.3f ms, by user %.3f ms (x%.1f); "
      "1M users = %d tiles: %.1f s -> %.1f s" % (t1, t2, t1 / t2, 1_000_000 // nu, t1 * 1953 / 1e3, t2 * 1953 / 1e3))
