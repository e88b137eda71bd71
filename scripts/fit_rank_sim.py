"""What ONE rank of an 8-GPU data-parallel fit computes per step, timed on one GPU, plus the bytes it puts on the wire.

The 1M x 1M, d = 128 WMRB workload of bench.py's fit leg; rank 0's share: users [0, U / 8), their interactions, identity user
features (columns [0, U / 8) of the 1M-column matrix), ALL items.  The exchange plan is the real one
(sharding.plan_gradient_exchange) evaluated for a world of 8 from rank 0's seat -- the other ranks' row supports are what
their shards of the identity matrix give -- and the collectives are replaced by their local halves (reduce-scatter -> this
rank's rows of its own gradient, all-gather -> nothing): every kernel a rank launches runs, at its real size, nothing moves.
Hardware numbers for the exchange itself need 8 GPUs; what is reported here is (a) per-rank compute per step, (b) bytes on
the wire per rank and step by table, (c) for orientation only, the time those bytes take at STATED link rates."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

W = int(os.environ.get("SIM_WORLD", "8"))
U, I, d, per_user, S = 1_000_000, 1_000_000, 128, 20, 100


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import tensorrec_amd as T
    from tensorrec_amd import sharding, ops
    sharding.FORCE_COLLECTIVES = True
    n_loc = U // W
    real_plan = sharding.plan_gradient_exchange

    def plan_for_world(names, shapes, supports, device, group=None):
        """rank 0's plan in a world of W: the other ranks' supports are their slices of the identity user features"""
        plan = sharding.GradPlan()
        for name in names:
            rows = int(shapes[name][0]) if len(shapes[name]) else 1
            numel = int(np.prod(shapes[name])) if len(shapes[name]) else 1
            sup = supports.get(name)
            if sup is not None and tuple(sup) == (0, n_loc) and rows == U:       # a user table: rank r touches rows [r U/W, (r+1) U/W)
                plan.mode[name] = "disjoint"
                plan.bounds[name] = [r * n_loc for r in range(W)] + [rows]
            elif numel >= sharding.SHARD_MIN_NUMEL and rows >= W:
                plan.mode[name] = "sharded"
                per = -(-rows // W)
                plan.bounds[name] = [min(r * per, rows) for r in range(W)] + [rows]
            else:
                plan.mode[name] = "replicated"
            if name in plan.bounds:
                plan.own[name] = (plan.bounds[name][0], plan.bounds[name][1])
        plan.key = tuple((n, plan.mode[n], tuple(plan.bounds.get(n, ()))) for n in names)
        return plan
    sharding.plan_gradient_exchange = plan_for_world
    sharding.reduce_scatter_rows = lambda grad, bounds, rank, group=None, async_op=False: (grad[bounds[0]:bounds[1]], None)
    sharding.all_gather_rows = lambda weights, bounds, rank, group=None, async_op=False: None
    sharding.sync_owned_rows = lambda tensors, bounds, group=None: None

    rng = np.random.default_rng(1000)
    cols = rng.integers(0, I, size=(n_loc, per_user), dtype=np.int32)
    inter = sp.csr_matrix((np.ones(n_loc * per_user, np.float32), cols.reshape(-1),
                           np.arange(0, (n_loc + 1) * per_user, per_user, dtype=np.int64)), shape=(n_loc, I))
    inter.sum_duplicates()
    inter.data[:] = 1.0
    uf = sp.csr_matrix((np.ones(n_loc, np.float32), np.arange(0, n_loc, dtype=np.int32), np.arange(n_loc + 1, dtype=np.int64)),
                       shape=(n_loc, U))
    itf = sp.identity(I, dtype=np.float32, format="csr")
    model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0, data_parallel=True,
                        dp_sync_every_call=False)

    def run(epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit_partial(inter, uf, itf, epochs=epochs, n_sampled_items=S, user_offset=0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(1)
    one = run(1)
    per_step = (run(6) - one) / 5.0
    ops.KERNEL_EVENTS = []
    run(2)
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dur = {}
    for n, a, b in ev:
        dur.setdefault(n, []).append(a.elapsed_time(b))
    plan = model._dp_plan
    shapes = {n: tuple(v.shape) for n, v in model._store.variables.items()}
    wire = plan.wire_bytes_per_step(shapes)
    total = float(sum(wire.values()))
    out = {"world_emulated": W, "per_rank_compute_ms_per_step": 1e3 * per_step,
           "ideal_strong_scaling_ms": None, "plan": dict(plan.mode),
           "wire_bytes_per_rank_and_step": wire, "wire_bytes_per_rank_and_step_total": total,
           "round3_wire_bytes_per_rank_and_step": 2.0 * (W - 1) / W * sum(float(np.prod(s)) * 4 for s in shapes.values()),
           "kernels_avg_ms": {n: float(np.mean(v)) for n, v in dur.items()},
           "exchange_ms_at_stated_rates_NOT_MEASURED": {
               "note": "bytes / rate; reduce-scatter and all-gather each move half of the total; RCCL bus bandwidth on an 8-GPU xGMI "
                       "mesh is not known here -- two brackets",
               "at_150_GBps_per_rank": 1e3 * total / 150e9, "at_350_GBps_per_rank": 1e3 * total / 350e9}}
    print(json.dumps(out))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fit_rank_sim.json")
    json.dump(out, open(path, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
