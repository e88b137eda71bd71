#!/usr/bin/env python3
"""
bench.py -- the headline measurement of BASELINE.json: user-item predictions/sec on the synthetic 1M-user x 1M-item,
identity-feature, d = 128 DotProduct (biased) model -- configs[2], the configuration the metric is quoted on.

One "step" = one full pass of the scoring hot path over ALL users, inputs (CSR features, weights) resident in HBM:
    K1 CSR gather-SpMM (user and item representations) + bias SpMVs -> operand preparation (bf16) ->
    K2 fused MFMA score + per-user top-10 (the [U, I] matrix is never written: it would be 4 TB) -> K5 merge.
value = U * I_total / step_time (every user-item pair is scored each step).

N GPUs: one process per GPU (torchrun, RCCL).  Items are sharded row-wise (strong scaling: the problem stays
1M x 1M), the user side is replicated, and each step ends with ONE all-gather of the per-shard top-10 lists followed
by the local merge (tensorrec_amd/sharding.py).

Extra objects on the JSON line: "roofline" (K2, MFMA-bound; K1's HBM roofline is under "roofline_k1"),
"cpu_baseline" (the oracle's NumPy/torch-CPU path on the host cores, on a bounded user tile), "parity" (a live check of
a sample of users against the oracle inside this run).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
INT8_DENSE_PEAK_TOPS = 5000.0       # MI355X_MICROARCH.md: I8 MFMA "~2x bf16 rate" (no spec row; its ubench ceiling is >= 3944 TOPS)
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
FIT_SHARD_USERS = 49152             # the oracle's large fit shard: 4.9M sampled pairs (>= 2^22: the binned route), ~30 s of CPU per step
GATHER_CEILING_512MB_GBS = 7420.0    # measured: bare 512-byte row gathers from a 512 MB table (profiles/r05_gather_ceiling_512.jsonl)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prewarm-seconds", type=float, default=0.0,
                    help="untimed steps run for about this long BEFORE the warmup steps (0 = none): the first GPU process on a fresh "
                         "box launches kernels ~40x slower for its first half minute (measured: 105.9 -> 101.7 -> 95.2 ms per step in "
                         "three consecutive processes), which says nothing about the steady state the metric is quoted for")
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--components", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--precision", default="exact", choices=["exact", "bf16", "fp32"],
                    help="exact (default): fp32 results, contraction on bf16 MFMA as an error-bounded filter + fp32 "
                         "re-scoring of the survivors; bf16: approximate bf16 scores; fp32: fp32 MFMA throughout")
    ap.add_argument("--prefilter", default="int8", choices=["int8", "none"],
                    help="exact mode: int8 MFMA pre-filter in front of the bf16 stage (the cascade of csrc/topk_cascade.hip)")
    ap.add_argument("--variant", type=int, default=int(os.environ.get("TREC_SCORE_VARIANT", "1")))
    ap.add_argument("--chunks", type=int, default=0, help="item chunks per user block (0 = auto)")
    ap.add_argument("--method", default="auto", choices=["auto", "direct", "two_stage"])
    ap.add_argument("--unbiased", action="store_true", help="diagnostic: model built with biased=False")
    ap.add_argument("--no-fit", action="store_true", help="skip the fit epochs/sec measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-users", type=int, default=4096,
                    help="user tile of the CPU baseline (SURVEY.md 8d: 4,096 users x all items)")
    ap.add_argument("--parity-users", type=int, default=4096, help="users checked against the oracle in exact mode")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the all-fp32-MFMA record")
    ap.add_argument("--no-k1-multi", action="store_true", help="skip the multi-nnz K1 roofline")
    ap.add_argument("--configs", default="all", choices=["all", "headline"],
                    help="all (default): also the records of BASELINE.json configs[1], [3] (one rank's shard) and [4], the "
                         "trained-weights record, parity_fit and the multi-nnz parity; headline: the 1M x 1M line only")
    ap.add_argument("--trained-epochs", type=int, default=20, help="epochs of the fit behind the trained-weights record")
    ap.add_argument("--fused-k1", action="store_true",
                    help="K1 emits the filter's operands in its epilogue (trec_spmm_csr_filter) instead of a separate prep "
                         "pass: 0.1 ms per side less in total, but the gather kernel itself then runs at 0.49 of the HBM "
                         "roofline instead of 0.68 (DESIGN.md section 8)")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher path only: every rank joins the process group, runs the collective self-check and rank 0 prints "
                         "one line -- no workload (runs without a GPU under TREC_DIST_BACKEND=gloo: tests/test_host_logic.py)")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=INT",
                    help="diagnostic: set a kernel tuning knob (trec_set_tuning), e.g. blockmax_pipelined=0")
    return ap.parse_args()


def csrc_stamp():
    """sha1 (12 hex) of every kernel source: the header of each profiles/*pmc_summary.txt carries the stamp of the code it was
    taken from (scripts/gpu_pmc_cmd.sh, scripts/gpu_profile.sh), and a counter is quoted only while the sources of its kernel
    are unchanged."""
    import glob, hashlib
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "tensorrec_amd", "csrc", "*.h*"))):
        out[os.path.basename(f)] = hashlib.sha1(open(f, "rb").read()).hexdigest()[:12]
    return out


def pmc_counters(file_glob, kernel_prefix, sources, counters=("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT", "TCC_MISS")):
    """Per-dispatch averages of `counters` for the first kernel whose name starts with `kernel_prefix`, from the newest committed
    summary matching `file_glob`.  Returns (values or None, file name or None, stale): stale = the summary carries no source stamp,
    or the stamp of one of `sources` (the files the kernel is compiled from) differs from the tree's -- the caller then reports
    traffic: null and traffic_stale: <file> instead of a number from other code."""
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", file_glob)))
    if not files:
        return None, None, False
    txt = open(files[-1]).read()
    name = os.path.basename(files[-1])
    m = re.search(r"^# csrc_sha: (\{.*\})$", txt, re.M)
    if not m:
        return None, name, True
    stamp, now = json.loads(m.group(1)), csrc_stamp()
    if any(stamp.get(f) != now.get(f) for f in tuple(sources) + ("common.hpp",)):
        return None, name, True
    vals = {}
    for c in counters:
        hit = re.findall(re.escape(kernel_prefix) + r"[^\n]*?\b" + c + r"=([0-9.e+]+)", txt)
        if hit:
            vals[c] = float(hit[0])
    return (vals or None), name, False


def attach_traffic(roof, file_glob, kernel_prefix, sources, note=""):
    """roof["traffic"] = (2*FETCH_SIZE + WRITE_SIZE) KB per launch (gfx950 counts wide coalesced reads at half their size:
    MI355X_MICROARCH.md, HBM section), or null + traffic_stale when the committed counters are of other code."""
    if roof is None:
        return
    vals, name, stale = pmc_counters(file_glob, kernel_prefix, sources)
    if stale:
        roof["traffic"], roof["traffic_stale"] = None, name
        return
    if not vals or "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return
    roof["traffic"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    roof["traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KB per launch from %s%s" % (name, note)
    if "TCC_HIT" in vals and "TCC_MISS" in vals:
        roof["l2_hit_rate"] = vals["TCC_HIT"] / (vals["TCC_HIT"] + vals["TCC_MISS"])


def cpu_baseline(n_items, d, k, n_users_sample, seed=0):
    """The oracle's CPU path for the same pass: scipy CSR @ (identity features), torch-CPU sgemm,
    bias broadcast, NumPy argpartition top-k; float32; all host cores.  Timed on a bounded user tile x ALL items."""
    import scipy.sparse as sp
    import torch
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rng = np.random.default_rng(seed)
    w_i = O.init_linear_weights(n_items, d, rng)
    w_u = O.init_linear_weights(n_users_sample, d, rng)
    f_i = sp.identity(n_items, dtype=np.float32, format="csr")
    f_u = sp.identity(n_users_sample, dtype=np.float32, format="csr")
    b_i = np.zeros((n_items, 1), np.float32)
    b_u = np.zeros((n_users_sample, 1), np.float32)
    t0 = time.perf_counter()
    item_repr = O.linear_repr(f_i, w_i)
    user_repr = O.linear_repr(f_u, w_u)
    ib = O.project_biases(f_i, b_i)
    ub = O.project_biases(f_u, b_u)
    it = torch.from_numpy(item_repr)
    block = 256
    for s in range(0, n_users_sample, block):
        scores = (torch.from_numpy(user_repr[s:s + block]) @ it.t()).numpy()
        scores = O.bias_prediction_dense(scores, ub[s:s + block], ib)
        part = np.argpartition(-scores, k, axis=1)[:, :k]
        top = np.take_along_axis(scores, part, axis=1)
        order = np.argsort(-top, axis=1, kind="stable")
        np.take_along_axis(part, order, axis=1)
    dt = time.perf_counter() - t0
    return {"value": n_users_sample * n_items / dt, "unit": "predictions/s", "cores": cores, "kind": "port",
            "sample": "oracle (scipy CSR + torch-CPU sgemm + numpy argpartition, fp32) on %d users x %d items, "
                      "d=%d, top-%d, %.1f s" % (n_users_sample, n_items, d, k, dt)}


def cpu_baseline_fit(n_users_total, n_items, d, per_user=20, n_sampled=100, shard=49152, small_shard=4096,
                     small_shard_seconds=None, shard_seconds=None, seed=0):
    """CPU leg of the fit half of the metric: ONE optimiser step of the oracle's model (oracle/model.py -- the restated
    _build_tf_graph + TF-form Adam, torch-CPU autograd, float32, all host cores) on a user shard of the same 1M-item, d = 128,
    WMRB workload.  Reported: the MEASURED rate of a 49,152-user shard, timed twice (both times given -- round 2's driver run
    moved 2-3x between runs), and, separately and labelled as such, the extrapolation to one step over all users: a step costs
    a + b * users (a: the item-side dense work -- 1M x 128 weights, their Adam update), a and b from the 4,096-user step
    that parity_fit times anyway and the faster of the two large-shard times.  `shard_seconds`: a step of this shard size the
    run has already timed (parity_fit_binned's oracle step, same workload) -- it then counts as the first of the two."""
    import scipy.sparse as sp
    import torch
    from oracle.model import OracleTensorRec
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    itf = sp.identity(n_items, dtype=np.float32, format="csr")

    def one(n_u):
        rng = np.random.default_rng(seed)
        cols = rng.integers(0, n_items, size=(n_u, per_user), dtype=np.int64)
        inter = sp.csr_matrix((np.ones(n_u * per_user, np.float32), cols.reshape(-1),
                               np.arange(0, (n_u + 1) * per_user, per_user, dtype=np.int64)), shape=(n_u, n_items))
        inter.sum_duplicates()
        uf = sp.identity(n_u, dtype=np.float32, format="csr")
        model = OracleTensorRec(d, "linear", "linear", "dot", "wmrb", True)
        model.init_weights(n_u, n_items, rng)
        samples = rng.integers(0, n_items, size=(n_u, n_sampled), dtype=np.int64)   # (cost only: distinctness is irrelevant)
        t0 = time.perf_counter()
        model.step(inter, uf, itf, 0.1, 1e-5, samples)
        return time.perf_counter() - t0
    if small_shard_seconds is None:
        one(64)                                           # the first, tiny step pays the one-time costs and is dropped
        small_shard_seconds = one(small_shard)
    big = [one(shard) if shard_seconds is None else float(shard_seconds), one(shard)]
    t2 = min(big)
    b = max(0.0, (t2 - small_shard_seconds) / float(shard - small_shard))
    a = max(0.0, small_shard_seconds - b * small_shard)
    epoch = a + b * n_users_total
    return {"value": 1.0 / epoch, "unit": "epochs/s", "cores": cores, "kind": "port",
            "measured_shard": {"users": shard, "seconds": big, "users_per_second": shard / t2,
                               "epoch_if_users_ran_in_shards_of_this_size_s": t2 * n_users_total / float(shard)},
            "extrapolation": {"small_shard_users": small_shard, "small_shard_seconds": small_shard_seconds, "fixed_s": a,
                              "per_user_s": b, "one_step_over_all_users_s": epoch,
                              "label": "EXTRAPOLATED (two measured points, a + b * users); `value` = 1 / this"},
            "sample": "oracle/model.py (torch-CPU autograd + NumPy TF-form Adam, fp32): one optimiser step on %d users (%.1f s) and "
                      "twice on %d users (%.1f / %.1f s) x %d items, d=%d, WMRB, %d interactions + %d samples per user -> step = "
                      "%.2f s + %.3g s/user, extrapolated to one step over %d users = %.1f s"
                      % (small_shard, small_shard_seconds, shard, big[0], big[1], n_items, d, per_user, n_sampled, a, b,
                         n_users_total, epoch)}


def fit_epochs_per_sec(n_users, n_items, d, per_user=20, n_sampled=100, epochs=3, rank=0, world=1):
    """Second half of BASELINE.json's metric: fit epochs/sec on the same 1M x 1M, d=128 shape.  One epoch = one
    optimiser step over all users (user_batch_size=None) through the public API: K1 fwd (user + item), K7 sampling,
    K3 over the interactions and the U*S sampled pairs, K6 WMRB fwd+bwd, backward gathers (K1 on transposed / grouped
    structures), K8 dense Adam on every weight.  20 uniform-random positive interactions per user.
    N ranks: data-parallel over users (the reference's batching axis) -- every rank takes a contiguous slice of user
    rows; per weight the exchange plan of TensorRec._dp_make_plan (sharding.plan_gradient_exchange): tables whose touched
    rows are rank-disjoint (the user tables under these identity features) are stepped by their owner with no exchange,
    shared tables reduce-scatter -> Adam on the owned rows -> all-gather, small ones all-reduce; the problem stays 1M x 1M
    (strong scaling) and the time is the slowest rank's."""
    import scipy.sparse as sp
    import torch
    import tensorrec_amd as T
    from tensorrec_amd import sharding
    u0, u1 = (n_users * rank) // world, (n_users * (rank + 1)) // world
    rng = np.random.default_rng(1000 + rank)
    n_loc = u1 - u0
    cols = rng.integers(0, n_items, size=(n_loc, per_user), dtype=np.int32)
    indptr = np.arange(0, (n_loc + 1) * per_user, per_user, dtype=np.int64)
    inter = sp.csr_matrix((np.ones(n_loc * per_user, np.float32), cols.reshape(-1), indptr), shape=(n_loc, n_items))
    inter.sum_duplicates()
    inter.data[:] = 1.0
    # this rank's rows of the identity user-feature matrix (all n_users feature columns stay: weights are replicated)
    uf = sp.csr_matrix((np.ones(n_loc, np.float32), np.arange(u0, u1, dtype=np.int32),
                        np.arange(n_loc + 1, dtype=np.int64)), shape=(n_loc, n_users))
    itf = sp.identity(n_items, dtype=np.float32, format="csr")
    model = T.TensorRec(n_components=d, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=0, data_parallel=world > 1)
    device = torch.device("cuda", torch.cuda.current_device())

    def run(n_epochs):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        model.fit_partial(inter, uf, itf, epochs=n_epochs, n_sampled_items=n_sampled, user_offset=u0)
        torch.cuda.synchronize()
        return sharding.max_over_ranks(time.perf_counter() - t0, device)

    run(1)                                                           # build + warm-up
    one = run(1)
    per_epoch = (run(1 + epochs) - one) / epochs                     # removes the per-call upload of the inputs
    nnz = sharding.all_reduce_scalar(int(inter.nnz), device)
    # ---- roofline of the dominant fit kernel: HIP events around every launch of two more epochs ----
    from tensorrec_amd import ops
    ops.KERNEL_EVENTS = []
    run(2)
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dur = {}
    for name, s_, e_ in events:
        dur.setdefault(name, []).append(s_.elapsed_time(e_))
    roofline_fit = None
    if "wmrb_fused_step" in dur:
        ms = float(np.mean(dur["wmrb_fused_step"]))
        pairs_s, pairs_p = float(n_loc) * n_sampled, float(inter.nnz)
        # item rows gathered once per pair; user rows read, dU written; per pair: sample id + coefficient + bucket rank
        # (samples) / item id + prediction + coefficient + loss (interactions)
        alg = (pairs_s + pairs_p) * d * 4 + 2.0 * n_loc * d * 4 + pairs_s * 12 + pairs_p * 16
        gbs = alg / (ms * 1e-3) / 1e9
        roofline_fit = {"kernel": "wmrb_user_fused_kernel (one-pass WMRB step, user side: every pair's item row gathered once)",
                        # ("hbm+mall": half of the 512 MB item table is served by the 256 MB Infinity Cache, so the algorithmic rate may
                        # exceed what HBM alone streams (~6.3 TB/s); priced against the 8 TB/s peak and the measured gather ceiling)
                        "bound": "hbm+mall", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "traffic": None, "avg_launch_ms": ms, "launches": len(dur["wmrb_fused_step"]),
                        "algorithmic_bytes_per_launch": alg,
                        "gather_ceiling": GATHER_CEILING_512MB_GBS, "frac_of_gather_ceiling": gbs / GATHER_CEILING_512MB_GBS,
                        "bound_note": "random 512-byte row gathers from a 512 MB item table: 16x the aggregate L2 (32 MB) "
                                      "and 2x the Infinity Cache, so the rows cross the memory-side fabric (TCC hit rate "
                                      "and FETCH_SIZE of this kernel: profiles/r*_fit_pmc_summary.txt); priced against "
                                      "the HBM peak.  gather_ceiling = what BARE 512-byte row gathers from a 512 MB table "
                                      "deliver on this chip (scripts/probe/gather_ceiling.hip, profiles/r05_gather_ceiling_512.jsonl: "
                                      "7.4 TB/s; 7.7 TB/s from a table resident in the Infinity Cache -- blocking the pairs for "
                                      "the cache cannot pay, DESIGN 8)",
                        "other_kernels_avg_ms": {n: float(np.mean(v)) for n, v in dur.items() if n != "wmrb_fused_step"},
                        "other_kernels_launches_per_epoch": {n: len(v) / 2.0 for n, v in dur.items() if n != "wmrb_fused_step"}}
        try:                                                         # (the pairs the item side still sorts and gathers: DESIGN 0 (5c))
            st = ops.LAST_FUSED_STATS
            if st.get("sampled_pairs"):
                roofline_fit["sampled_pairs_kept_fraction"] = int(st["sampled_pairs_kept"].item()) / float(st["sampled_pairs"])
        except Exception:
            pass
        if world == 1 and (n_users, n_items, d) == (1_000_000, 1_000_000, 128):
            attach_traffic(roofline_fit, "r[0-9][0-9]_fit_pmc_summary.txt", "wmrb_user_fused_kernel", ("wmrb_fused.hip",))
    return {"fit_epochs_per_sec": 1.0 / per_epoch, "sec_per_epoch": per_epoch, "epochs_timed": epochs,
            "roofline_fit": roofline_fit,
            "workload": "%d users x %d items, identity features, d=%d, Linear + DotProduct + WMRB, biased, %d "
                        "interactions, n_sampled_items=%d, device sampler, 1 optimiser step per epoch"
                        % (n_users, n_items, d, nnz, n_sampled),
            "parallelism": "single GPU" if world == 1 else
                           "users sharded x%d (data-parallel); gradient exchange per table: %s" % (
                               world, ", ".join("%s=%s" % kv for kv in sorted((getattr(model, "_dp_plan", None) or
                                                                               sharding.GradPlan()).mode.items()))),
            "per_call_input_upload_sec": one - per_epoch}


def launch_ranks(n_ranks, argv, n_devices, env=None):
    """`python bench.py --gpus N` without a launcher around it: re-run this file as N ranks under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1 at a free port) and hand back its exit code.  The ranks speak RCCL, so N GPUs must
    be visible; with fewer the call is refused -- unless TREC_DIST_BACKEND=gloo asks for the functional form in which ranks
    share a GPU (tests, one-GPU boxes: never a reported number).  Rank 0 alone prints the JSON line (main())."""
    import socket
    import subprocess
    env = dict(os.environ if env is None else env)
    backend = env.get("TREC_DIST_BACKEND", "nccl")
    if backend == "nccl" and n_devices < n_ranks:
        raise SystemExit("--gpus %d: %d GPU(s) visible and RCCL needs one per rank (TREC_DIST_BACKEND=gloo runs the ranks "
                         "functionally on fewer GPUs)" % (n_ranks, n_devices))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC: the only form the host driver supports)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def launch_check(world, rank):
    """--launch-check: what a rank does to prove the launcher path without the workload -- join the process group, run
    sharding.collective_selfcheck on this rank's device (CPU tensors under gloo without a GPU), rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    from tensorrec_amd import sharding
    backend = os.environ.get("TREC_DIST_BACKEND", "nccl")
    if torch.cuda.is_available():
        dev_index = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    else:
        device = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    check = sharding.collective_selfcheck(device) if world > 1 else "single"
    total = sharding.all_reduce_scalar(rank + 1, device)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": check, "n_gpus": world, "backend": backend if world > 1 else None,
                          "rank_sum": total, "device": device.type}))
        sys.stdout.flush()
    return 0 if check in ("ok", "single") and total == world * (world + 1) // 2 else 1


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started bare (the driver's `python bench.py --gpus N`): become the launcher of N ranks of this same command
        import torch
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], torch.cuda.device_count()))
    if world != args.gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d: start the ranks with --nproc-per-node %d (or "
                         "run `python bench.py --gpus %d`, which launches them itself)" % (args.gpus, world, args.gpus, args.gpus))
    if args.launch_check:
        raise SystemExit(launch_check(world, rank))
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp

    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(1, n_dev)          # ranks share a GPU only in the single-GPU functional test below
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") needs one GPU per rank; TREC_DIST_BACKEND=gloo lets the multi-rank path be exercised
        # functionally on a one-GPU box (tests / gpurun), it is never used for a reported number
        backend = os.environ.get("TREC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import tensorrec_amd as T
    from tensorrec_amd import ops, sharding
    from tensorrec_amd.sparse import SparseFeatures

    for kv in args.tune:
        name, _, val = kv.partition("=")
        T._native.set_tuning(name, int(val))
    U, I, d, k = args.users, args.items, args.components, args.k
    dtype = ops.DTYPE_F32 if args.precision == "fp32" else ops.DTYPE_BF16     # arithmetic type of the MFMA stage
    exact = args.precision == "exact"
    i_begin, i_end = sharding.shard_bounds(I, world, rank, align=64)
    n_local = i_end - i_begin

    # ---- synthetic model: identity features, Linear init (N(0,1) rows, L2-normalised), zero biases (untrained) ----
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    w_u = torch.randn((U, d), device=device, generator=gen)
    w_i_full_seeded = torch.Generator(device=device)
    w_i_full_seeded.manual_seed(1)
    w_i = torch.randn((I, d), device=device, generator=w_i_full_seeded)[i_begin:i_end].contiguous()
    w_u = ops.l2_normalize_rows(w_u)
    w_i = ops.l2_normalize_rows(w_i)
    # per-feature biases: small and NON-zero (a freshly initialised model has zeros, any fitted one does not: the integer
    # item-bias path of the int8 stage and the (s + b_u) + b_i order are on the timed path and in the parity check)
    beta_u = 0.05 * torch.randn((U, 1), device=device, generator=gen)
    beta_i = (0.05 * torch.randn((I, 1), device=device, generator=w_i_full_seeded))[i_begin:i_end].contiguous()
    f_u = SparseFeatures(sp.identity(U, dtype=np.float32, format="csr"), device)
    f_i = SparseFeatures(sp.identity(n_local, dtype=np.float32, format="csr"), device)   # this rank's item rows
    kpad = ops.score_kpad(d)
    method = args.method
    if method == "auto":
        method = "two_stage" if n_local >= ops.TWO_STAGE_MIN_ITEMS else "direct"
    n_chunks = args.chunks if args.chunks > 0 else ops.topk_chunks_for(U, dtype, kpad, n_local)
    cap = T._native.query("trec_score_topk_capacity", k)
    n_parts = T._native.query("trec_score_topk_parts", dtype, kpad, n_local, n_chunks)
    ws = (torch.empty((U, n_parts, cap), dtype=torch.float32, device=device),
          torch.empty((U, n_parts, cap), dtype=torch.int32, device=device))

    # item shards exchange per-user data partitioned by user (all-to-all: every rank receives 1/world of what an
    # all-gather would deliver); the all-gather forms remain for backends without device all-to-all (gloo tests)
    selfcheck = None
    floor_fn, topk_fn = sharding.shared_topk_floor, sharding.sharded_top_k
    if world > 1:
        selfcheck = sharding.collective_selfcheck(device)
        if sharding.a2a_available(w_u) and selfcheck == "ok":
            floor_fn, topk_fn = sharding.shared_topk_floor_a2a, sharding.sharded_top_k_a2a

    prefilter = "int8" if exact and args.prefilter == "int8" and kpad in (64, 128) else None

    def make_step(w_u, w_i, beta_u, beta_i):
        def step():
            with torch.no_grad():
                if exact and method == "two_stage" and d in (32, 64, 128, 256) and args.fused_k1:
                    # K1 emits the filter's operands itself (fp32 representation + bf16 image + error norms): no prep pass
                    ub = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)
                    ib = None if args.unbiased else ops.sparse_matvec(f_i, beta_i)
                    u_f = ops.spmm_filter_operand(f_u, w_u)
                    i_f = ops.spmm_filter_operand(f_i, w_i, bias=ib, want_gstats=True)
                    vals, idx = ops.score_topk_filtered(
                        u_f, i_f, k, ub, ib, item_index_base=i_begin, variant=args.variant,
                        n_chunks=args.chunks if args.chunks > 0 else None,
                        floor_exchange=floor_fn if world > 1 else None,
                        stats_exchange=sharding.all_reduce_max if world > 1 else None, prefilter=prefilter)
                    if world > 1:
                        vals, idx = topk_fn(vals, idx, k)
                    return vals, idx, u_f.f32, i_f.f32
                user_repr = ops.spmm_raw(f_u.indptr, f_u.indices, f_u.values, None, U, f_u.nnz, w_u,          # K1
                                         one_per_row=f_u.one_per_row)
                item_repr = ops.spmm_raw(f_i.indptr, f_i.indices, f_i.values, None, n_local, f_i.nnz, w_i,    # K1
                                         one_per_row=f_i.one_per_row)
                ub = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)
                ib = None if args.unbiased else ops.sparse_matvec(f_i, beta_i)
                if exact and method == "two_stage":
                    # fp32-exact top-k: bf16 MFMA stage 1 as an error-bounded filter, survivors re-scored in fp32
                    # the user side in five launches, no host read: class sort + fp32 / bf16 / int8 operands + bias (csrc/user_prep.hip)
                    u_f = ops.score_prep_filter(user_repr, sort_users=prefilter == "int8", k=k, user_bias=ub)
                    i_f = ops.score_prep_filter(item_repr, bias=ib, want_gstats=True)
                    vals, idx = ops.score_topk_filtered(
                        u_f, i_f, k, ub, ib, item_index_base=i_begin, variant=args.variant,
                        n_chunks=args.chunks if args.chunks > 0 else None,
                        floor_exchange=floor_fn if world > 1 else None,
                        stats_exchange=sharding.all_reduce_max if world > 1 else None, prefilter=prefilter,
                        finish_lanes=16 if world >= 4 else 0)          # (short per-shard lists: four users per wave)
                    if world > 1:
                        vals, idx = topk_fn(vals, idx, k)              # every rank finalises ITS users (all-to-all + merge)
                    return vals, idx, user_repr, item_repr
                u_op, _, _ = ops.score_prep(user_repr, dtype)
                i_op, _, _ = ops.score_prep(item_repr, dtype)
                if method == "direct":
                    vals, idx = ops.score_topk_direct(u_op, i_op, dtype, kpad, k, ub, ib, item_index_base=i_begin,
                                                      n_chunks=n_chunks, variant=args.variant, workspace=ws)  # K2 + merge
                else:
                    # item shards share ONE top-k floor per user (all-gather of k superblock maxima) before re-scoring
                    vals, idx = ops.score_topk_two_stage(u_op, i_op, dtype, kpad, k, ub, ib, item_index_base=i_begin,
                                                         variant=args.variant, n_chunks=args.chunks if args.chunks > 0 else None,
                                                         floor_exchange=floor_fn if world > 1 else None)
                if world > 1:
                    vals, idx = topk_fn(vals, idx, k)
                return vals, idx, user_repr, item_repr
        return step

    step = make_step(w_u, w_i, beta_u, beta_i)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prewarm_steps = 0
    if args.prewarm_seconds > 0:
        step()                                       # (allocations, code objects)
        sync()
        t0 = time.perf_counter()
        step()
        sync()
        # every rank runs the same number of steps (the step holds collectives): the largest estimate of any rank
        prewarm_steps = int(sharding.max_over_ranks(min(500.0, args.prewarm_seconds / max(1e-3, time.perf_counter() - t0)), device))
        for _ in range(prewarm_steps):
            out = step()
    # (the results are HELD across the next step, exactly as in the timed loop below: the caching allocator then already owns the
    # second set of result buffers -- otherwise the second timed step is the first to run while a previous result is alive, and
    # pays the device allocations: +25 ms on that one step in every record up to round 6)
    out = None
    if args.warmup < 2:                              # (at least two untimed steps ever run before the clock starts: code objects, both result sets)
        for _ in range(2 - args.warmup):
            out = step()
            prewarm_steps += 1
    for _ in range(args.warmup):
        out = step()
    ops.KERNEL_EVENTS = []
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # step boundaries on the stream (no sync)
    t0 = time.perf_counter()
    marks[0].record()
    for i_ in range(args.steps):
        out = step()
        marks[i_ + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(args.steps)]
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    elapsed = sharding.max_over_ranks(elapsed, device)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = float(U) * float(I) / (elapsed / args.steps)

    if exact and prefilter == "int8":
        ops.CANDIDATE_STATS = True                   # one untimed step with the list statistics on (a reduction + a host read)
        out = step()
        ops.CANDIDATE_STATS = False
        torch.cuda.synchronize()
    if world > 1:
        # parity sample needs every rank's item rows: one untimed all-gather (shards are padded to equal length)
        vals, idx, user_repr, item_repr_local = out
        per = -(-I // world)
        per = -(-per // 64) * 64
        pad = torch.zeros((per, item_repr_local.shape[1]), dtype=item_repr_local.dtype, device=device)
        pad[: item_repr_local.shape[0]] = item_repr_local
        full = sharding.all_gather_cat(pad, dim=0)[:I].contiguous()
        out = (vals, idx, user_repr, full)
        padb = torch.zeros((per,), dtype=torch.float32, device=device)
        if not args.unbiased:
            padb[: n_local] = ops.sparse_matvec(f_i, beta_i).reshape(-1)
        item_bias_full = sharding.all_gather_cat(padb, dim=0)[:I].contiguous()
    fit = None
    if not args.no_fit:
        try:                      # every rank takes part (data-parallel over users for world > 1)
            del ws
            torch.cuda.empty_cache()
            fit = fit_epochs_per_sec(U, I, d, rank=rank, world=world)
        except Exception as exc:
            fit = {"error": repr(exc)}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- per-kernel durations from HIP events recorded on the launching stream during the timed region ----
    dur = {}
    for name, s, e in events:
        dur.setdefault(name, []).append(s.elapsed_time(e))
    cascade = "score_gemm_blockmax_i8" in dur and "score_gemm_blockmax" not in dur
    k2_name = "score_gemm_topk" if method == "direct" else ("score_gemm_blockmax_i8" if cascade else "score_gemm_blockmax")
    k2_ms = float(np.mean(dur[k2_name]))
    # launches of the dominant kernel per step: 1, or the user batches of the two-stream pipeline (ops.cascade_user_batches) --
    # a launch then covers U / batches users and runs NEXT TO the previous batch's bf16 refinement and finish
    k2_lps = max(1.0, len(dur[k2_name]) / float(args.steps))
    k2_flops_step = 2.0 * U * n_local * kpad               # algorithmic: 2*U*I*d per step (d = kpad = 128 here)
    k2_flops = k2_flops_step / k2_lps
    per_step = lambda v: float(np.sum(v)) / float(args.steps)
    peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else (INT8_DENSE_PEAK_TOPS if cascade else BF16_DENSE_PEAK_TFLOPS)
    k2_tflops = k2_flops / (k2_ms * 1e-3) / 1e12
    k2_label = "score_gemm_kernel (fused top-k epilogue)" if method == "direct" else (
        "blockmax_i8x16_kernel (int8 MFMA, all pairs; peak = 2x bf16 dense)" if cascade else
        "blockmax_pipe_kernel (superblock maxima, stage 1 of the two-stage top-k)"
        if args.precision != "fp32" and kpad in (64, 128) and T._native.load().trec_get_tuning(b"blockmax_pipelined", 1)
        else "score_gemm_kernel (superblock-max epilogue)")
    roofline_bf16_stage = None
    if cascade and "score_gemm_blockmax_grouped" in dur:
        # (superblock, user) pairs the bf16 kernel works on: the pre-refinement's k per user (its own launch) + the compaction's
        rows = float(ops.LAST_FILTER_STATS.get("refined_rows", 0)) + float(ops.LAST_FILTER_STATS.get("prerefined_pairs", 0))
        g_ms = per_step(dur["score_gemm_blockmax_grouped"]) + per_step(dur.get("score_gemm_blockmax_pre", [0.0]))
        g_tf = 2.0 * rows * ops.SUPERBLOCK_ROWS * kpad / (g_ms * 1e-3) / 1e12
        roofline_bf16_stage = {"kernel": "blockmax_bf16x16_kernel, grouped form (v_mfma_f32_16x16x32_bf16: bf16 maxima of the "
                                         "(superblock, user) pairs the int8 bound cannot rule out)", "bound": "mfma", "achieved": g_tf,
                               "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g_tf / BF16_DENSE_PEAK_TFLOPS,
                               "avg_launch_ms": g_ms, "resident_rows": rows,
                               "prerefine_launch_ms": per_step(dur.get("score_gemm_blockmax_pre", [0.0])),
                               "prerefined_pairs": float(ops.LAST_FILTER_STATS.get("prerefined_pairs", 0)),
                               "refined_fraction_of_pairs": rows / (float(U) * ((n_local + ops.SUPERBLOCK_ROWS - 1) // ops.SUPERBLOCK_ROWS))}
    roofline = {"kernel": k2_label,
                "bound": "mfma", "achieved": k2_tflops,
                "peak": peak, "unit": "TFLOP/s", "frac": k2_tflops / peak, "traffic": None,
                "avg_launch_ms": k2_ms, "launches": len(dur[k2_name]), "launches_per_step": k2_lps,
                "algorithmic_flops_per_launch": k2_flops,
                "other_kernels_avg_ms": {n: per_step(v) for n, v in dur.items() if n not in (k2_name, "spmm_csr")},
                "other_kernels_note": "HIP-event time per step, summed over a step's launches of each kernel"
                                      + ("; with user batches these launches overlap the next batch's int8 launch on a second stream, "
                                         "so the sum of all kernels exceeds the step time" if k2_lps > 1 else "")}
    k1 = dur.get("spmm_csr", [])
    roofline_k1 = None
    if k1:
        # user-side launches are the even ones (U rows), item-side the odd ones (n_local rows)
        k1_user_ms = float(np.mean(k1[0::2]))
        # idx int32 + val, gather, store (+ the int64 row pointer unless every row has exactly one non-zero: not read then)
        bytes_user = U * (4 + 4) + (0 if f_u.one_per_row else (U + 1) * 8) + U * d * 4 + U * d * 4
        fused_k1 = exact and method == "two_stage" and d in (32, 64, 128, 256) and args.fused_k1
        if fused_k1:
            bytes_user += U * d * 2 + U * 8                                 # + the bf16 image and the two norms per row
        gbs = bytes_user / (k1_user_ms * 1e-3) / 1e9
        roofline_k1 = {"kernel": "%s (user side, identity features%s)"
                                 % ("spmm_one_per_row_kernel" if f_u.one_per_row and not fused_k1 else "spmm_csr_vec4_kernel",
                                    ", filter-operand epilogue: fp32 + bf16 rows + error norms" if fused_k1 else ""), "bound": "hbm",
                       "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                       "traffic": None, "avg_launch_ms": k1_user_ms, "algorithmic_bytes_per_launch": bytes_user}

    # ---- K1 on features that are NOT the identity: 20 non-zeros per row over 1M feature columns, d = 128 -- the 512 MB
    # weight table is 16x the L2 and 2x the Infinity Cache, so (unlike a small table) every gathered row is memory
    # traffic and nnz * d * 4 is an honest algorithmic byte count (checked against PMC: profiles/*k1_multi_pmc*.txt)
    roofline_k1_multi = None
    if world == 1 and not args.no_k1_multi:
        try:
            nnz_row, F = 20, 1_000_000
            n_rows = min(U, 1_000_000)
            rng = np.random.default_rng(5)
            cols = rng.integers(0, F, size=(n_rows, nnz_row), dtype=np.int32)
            cols.sort(axis=1)
            m = sp.csr_matrix((rng.random(n_rows * nnz_row, dtype=np.float32), cols.reshape(-1),
                               np.arange(0, (n_rows + 1) * nnz_row, nnz_row, dtype=np.int64)), shape=(n_rows, F))
            f_m = SparseFeatures(m, device)
            w_m = torch.randn((F, d), device=device, generator=gen)
            ops.spmm_raw(f_m.indptr, f_m.indices, f_m.values, None, n_rows, f_m.nnz, w_m)
            ops.KERNEL_EVENTS = []
            for _ in range(5):
                ops.spmm_raw(f_m.indptr, f_m.indices, f_m.values, None, n_rows, f_m.nnz, w_m)
            torch.cuda.synchronize()
            ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            ms = float(np.mean([s_.elapsed_time(e_) for n_, s_, e_ in ev if n_ == "spmm_csr"]))
            alg = f_m.nnz * (4 + 4) + (n_rows + 1) * 8 + float(f_m.nnz) * d * 4 + float(n_rows) * d * 4
            gbs = alg / (ms * 1e-3) / 1e9
            roofline_k1_multi = {"kernel": "spmm_csr_vec4_kernel, %d rows x %d non-zeros over %d feature columns, d=%d" % (n_rows, nnz_row, F, d),
                                 "bound": "hbm+mall", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "bound_note": "about half of the 512 MB weight table is served by the 256 MB Infinity Cache (MALL): the "
                                               "fabric-side counters include those hits, so this is NOT an HBM-only fraction (the HBM-only "
                                               "copy ceiling of the guide is ~6.3 TB/s)",
                                 "traffic": None, "avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg}
            if n_rows == 1_000_000 and d == 128:
                attach_traffic(roofline_k1_multi, "r[0-9][0-9]_k1_multi_pmc_summary.txt", "void spmm_csr_vec4_kernel", ("spmm.hip",))
            del f_m, w_m, m, cols
        except Exception as exc:
            roofline_k1_multi = {"error": repr(exc)}

    # ---- HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure
    # comes from the committed rocprofv3 --pmc passes of this same command (profiles/*_pmc_summary.txt), corrected as
    # MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts wide coalesced reads at half their size) ----
    if world == 1 and (U, I, d) == (1_000_000, 1_000_000, 128) and method == "two_stage":
        alg_note = "; fabric-side reads incl. Infinity-Cache hits; algorithmic minimum is %.3g bytes" % (
            (U + n_local) * kpad * (1.0 if cascade else 2.0) + U * 4.0 * (n_local // 512))
        if cascade:
            attach_traffic(roofline, "r[0-9][0-9]_pmc_summary.txt", "void (anonymous namespace)::blockmax_i8x16_kernel<128",
                           ("score_blockmax_i8.hip", "score_common.hpp"), alg_note)
        else:
            attach_traffic(roofline, "r[0-9][0-9]_pmc_summary.txt", "void (anonymous namespace)::blockmax_bf16x16_kernel<128",
                           ("score_blockmax.hip", "score_common.hpp"), alg_note)
        attach_traffic(roofline_k1, "r[0-9][0-9]_pmc_summary.txt",
                       "void spmm_one_per_row_kernel" if f_u.one_per_row else "void spmm_csr_vec4_kernel<1, 4, 0, true, true", ("spmm.hip",))

    # ---- live parity check against the oracle (checker only): the timed step's own output, on sampled users ----
    parity = None
    try:
        from oracle import oracle as O
        import bench_records as BR
        vals, idx, user_repr, item_repr = out
        n_sample = args.parity_users if exact else 32
        n_have = int(vals.shape[0])               # rank 0 holds all users (N = 1) or its own slice [0, n_have) (N > 1)
        sample = np.unique(np.linspace(0, n_have - 1, min(n_sample, n_have)).astype(np.int64))
        sample_dev = torch.from_numpy(sample).to(device)
        got_i = idx[sample_dev].cpu().numpy()
        got_v = vals[sample_dev].cpu().numpy()
        if world == 1 and not args.unbiased:
            # the oracle starts from the WEIGHTS: its own SpMM (tr_oracle.c) over the feature rows of the sampled users and of
            # every item, its own bias projection (non-zero biases), its fp32 score chain, its top-k
            fs = BR._identity_rows(sample, U)
            fi = sp.identity(I, dtype=np.float32, format="csr")
            us_all = O.spmm_exact(fs, w_u.cpu().numpy())
            it = O.spmm_exact(fi, w_i.cpu().numpy())
            ub_h = O.spmm_exact(fs, beta_u.cpu().numpy()).reshape(-1)
            ib_h = O.spmm_exact(fi, beta_i.cpu().numpy()).reshape(-1)
            source = "oracle SpMM + bias projection from the weights (non-zero biases)"
        else:
            # item shards: rank 0 does not hold the other ranks' weights; the gathered representations feed the oracle
            us_all, it = user_repr[sample_dev].cpu().numpy(), item_repr.cpu().numpy()
            ub_h = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)[sample_dev].cpu().numpy()
            ib_h = None if args.unbiased else item_bias_full.cpu().numpy()
            source = "gathered device representations and biases"
        parity = BR.oracle_topk_parity(O, us_all, it, ub_h, ib_h, got_v, got_i, k)
        parity["mode"] = args.precision
        parity["oracle_inputs"] = source
        if exact:
            parity["filter"] = dict(ops.LAST_FILTER_STATS)
    except Exception as exc:      # the measurement stands on its own; report why the check could not run
        parity = {"error": repr(exc)}

    # ---- the all-fp32-MFMA form of the same top-k (v_mfma_f32_32x32x2_f32 throughout), as a driver-run record ----
    fp32_mode = None
    if exact and world == 1 and not args.no_fp32_mode:
        try:
            vals, idx, user_repr, item_repr = out
            n32 = min(U, 65536)
            u32, _, _ = ops.score_prep(user_repr[:n32].contiguous(), ops.DTYPE_F32)
            i32, _, _ = ops.score_prep(item_repr, ops.DTYPE_F32)
            ub32 = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)[:n32].contiguous()
            ib32 = None if args.unbiased else ops.sparse_matvec(f_i, beta_i)
            ops.score_topk_two_stage(u32, i32, ops.DTYPE_F32, kpad, k, ub32, ib32)        # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                ev, ei = ops.score_topk_two_stage(u32, i32, ops.DTYPE_F32, kpad, k, ub32, ib32)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            # ... and the equality check over ALL users (the fp32 MFMA path is bit-exact against the oracle: tests), untimed
            all_equal = bool(torch.equal(ei, idx[:n32]) and torch.equal(ev, vals[:n32]))
            ub_all = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)
            for s0 in range(n32, U, n32):
                if not all_equal:
                    break
                uu, _, _ = ops.score_prep(user_repr[s0:s0 + n32].contiguous(), ops.DTYPE_F32)
                cv_, ci_ = ops.score_topk_two_stage(uu, i32, ops.DTYPE_F32, kpad, k,
                                                    None if ub_all is None else ub_all[s0:s0 + n32].contiguous(), ib32)
                all_equal = bool(torch.equal(ci_, idx[s0:s0 + n32]) and torch.equal(cv_, vals[s0:s0 + n32]))
            fp32_mode = {"workload": "%d users x %d items, whole two-stage top-%d on fp32 MFMA" % (n32, n_local, k),
                         "equals_timed_exact_mode_output_all_%d_users" % U: all_equal, "equals_all_users": all_equal,
                         "ms": 1e3 * dt, "predictions_per_s": n32 * float(n_local) / dt,
                         "tflops": 2.0 * n32 * n_local * kpad / dt / 1e12,
                         "frac_of_fp32_mfma_peak": 2.0 * n32 * n_local * kpad / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "equals_timed_exact_mode_output": bool(torch.equal(ei, idx[:n32]) and torch.equal(ev, vals[:n32]))}
        except Exception as exc:
            fp32_mode = {"error": repr(exc)}

    # ---- the bf16-filter form of the same exact top-k (no int8 stage): north_star's bf16 MFMA score kernel, dense, as a
    # driver-run record beside the cascade that is timed above ----
    bf16_mode = None
    if cascade and world == 1 and not args.no_fp32_mode:
        try:
            vals, idx, user_repr, item_repr = out
            ub2 = None if args.unbiased else ops.sparse_matvec(f_u, beta_u)
            ib2 = None if args.unbiased else ops.sparse_matvec(f_i, beta_i)

            def bf16_step():
                u_f = ops.score_prep_filter(user_repr)
                i_f = ops.score_prep_filter(item_repr, bias=ib2, want_gstats=True)
                return ops.score_topk_filtered(u_f, i_f, k, ub2, ib2, item_index_base=i_begin, variant=args.variant)
            bf16_step()
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS = []
            t0 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                bv, bi_ = bf16_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            ev2, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            s1 = float(np.mean([a.elapsed_time(b) for n_, a, b in ev2 if n_ == "score_gemm_blockmax"]))
            bf16_mode = {"workload": "the same %d users x %d items, exact top-%d through the bf16 filter alone (--prefilter none), "
                                     "operands given" % (U, n_local, k),
                         "ms": 1e3 * dt, "stage1_kernel": "blockmax_bf16x16_kernel (v_mfma_f32_16x16x32_bf16, dense bf16 superblock maxima)",
                         "stage1_avg_launch_ms": s1, "stage1_tflops": k2_flops_step / (s1 * 1e-3) / 1e12,
                         "stage1_frac_of_bf16_mfma_peak": k2_flops_step / (s1 * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                         "equals_timed_cascade_output": bool(torch.equal(bi_, idx) and torch.equal(bv, vals))}
        except Exception as exc:
            ops.KERNEL_EVENTS = None
            bf16_mode = {"error": repr(exc)}
    # north_star's own target kernel -- the DENSE bf16 MFMA score kernel -- as a roofline object of its own: time from the HIP
    # events of the record above, HBM-side traffic from the committed rocprofv3 PMC passes of `bench.py --prefilter none`
    roofline_bf16_dense = None
    if bf16_mode is not None and "error" not in bf16_mode:
        tf_ = bf16_mode["stage1_tflops"]
        roofline_bf16_dense = {"kernel": "blockmax_bf16x16_kernel<128, true, false> (v_mfma_f32_16x16x32_bf16: dense bf16 superblock "
                                         "maxima of every (user, item) pair -- stage 1 of the bf16 filter, north_star's score kernel)",
                               "bound": "mfma", "achieved": tf_, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf_ / BF16_DENSE_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_ms": bf16_mode["stage1_avg_launch_ms"], "algorithmic_flops_per_launch": k2_flops_step}
        if (U, I, d) == (1_000_000, 1_000_000, 128):
            attach_traffic(roofline_bf16_dense, "r[0-9][0-9]_bf16dense_pmc_summary.txt",
                           "void (anonymous namespace)::blockmax_bf16x16_kernel<128, true, false", ("score_blockmax.hip", "score_common.hpp"),
                           "; algorithmic minimum %.3g bytes (bf16 operands once + the [n_sb, U] table)"
                           % ((U + n_local) * kpad * 2.0 + U * 4.0 * (n_local // 512)))

    # ---- the PUBLIC API on the same weights: TensorRec.predict_top_k itself (features given as scipy matrices, uploaded once
    # and recognised by content afterwards; representations, user-side preparation, cascade; lists returned on the device), with
    # its default user batch -- the documented drop-in call next to the ops-level step that is timed above
    public_api = None
    if exact and world == 1 and not args.no_fp32_mode:
        try:
            vals, idx, user_repr, item_repr = out
            model = T.TensorRec(n_components=d, biased=not args.unbiased, seed=0)
            model.build(U, I)
            with torch.no_grad():
                model._store.variables["linear_weights_user_0"].copy_(w_u)
                model._store.variables["linear_weights_item"].copy_(w_i)
                if not args.unbiased:
                    model._store.variables["user_feature_biases"].copy_(beta_u)
                    model._store.variables["item_feature_biases"].copy_(beta_i)
            uf_sp = sp.identity(U, dtype=np.float32, format="csr")
            if_sp = sp.identity(I, dtype=np.float32, format="csr")
            model.predict_top_k(uf_sp, if_sp, k=k, return_device=True)            # first call: uploads, allocations
            torch.cuda.synchronize()
            reps, times = 3, []
            for _ in range(reps):
                t0 = time.perf_counter()
                pv, pi_ = model.predict_top_k(uf_sp, if_sp, k=k, return_device=True)
                torch.cuda.synchronize()
                times.append(1e3 * (time.perf_counter() - t0))
            t0 = time.perf_counter()
            hv, hi_ = model.predict_top_k(uf_sp, if_sp, k=k)                      # ... and with the lists copied to the host
            host_ms = 1e3 * (time.perf_counter() - t0)
            public_api = {"call": "TensorRec.predict_top_k(user_features, item_features, k=%d, return_device=True), default "
                                  "user_batch_size (= %d users per pass from the free device memory)"
                                  % (k, ops.topk_user_batch(U, I, d, device)),
                          "ms_per_call": times, "ms_per_call_min": min(times),
                          "predictions_per_s": float(U) * float(I) / (min(times) * 1e-3),
                          "ratio_to_timed_step": min(times) / ms_per_step,
                          "ms_per_call_with_host_copy_of_the_lists": host_ms,
                          "equals_timed_step_output": bool(torch.equal(pi_, idx) and torch.equal(pv, vals)),
                          "filter": dict(ops.LAST_FILTER_STATS)}
            del model, pv, pi_, hv, hi_
            torch.cuda.empty_cache()
        except Exception as exc:
            public_api = {"error": repr(exc)}

    # ---- further driver-visible records (rank 0, one GPU): fitted weights, fit parity, multi-nnz parity, the other configs
    trained = parity_fit = parity_fit_binned = parity_multi = configs = None
    oracle_small_shard_s = oracle_shard_s = None
    if world == 1 and args.configs == "all" and exact:
        import bench_records as BR
        try:
            trained = BR.trained_weights_record(make_step, device, U, I, d, k, steps=max(2, min(args.steps, 3)),
                                                epochs=args.trained_epochs)
        except Exception as exc:
            trained = {"error": repr(exc)}
        torch.cuda.empty_cache()
        try:
            vals, idx, user_repr, item_repr = out
            parity_multi = BR.parity_multi_nnz_record(device, item_repr, None if args.unbiased else ops.sparse_matvec(f_i, beta_i),
                                                      d, k)
        except Exception as exc:
            parity_multi = {"error": repr(exc)}
        if not args.no_cpu_baseline:
            try:
                parity_fit, oracle_small_shard_s = BR.parity_fit_record(I, d)
            except Exception as exc:
                parity_fit = {"error": repr(exc)}
            try:
                # the route the 1M x 1M fit itself takes (>= 2^22 sampled pairs: rank-free binned grouping, zero coefficients
                # dropped) against the oracle -- the 49,152-user oracle step is the one cpu_baseline_fit times anyway
                torch.cuda.empty_cache()
                parity_fit_binned, oracle_shard_s = BR.parity_fit_record(I, d, n_users=FIT_SHARD_USERS, expect_route="binned")
            except Exception as exc:
                parity_fit_binned = {"error": repr(exc)}
        torch.cuda.empty_cache()
        configs = BR.config_records(device)

    cpu = cpu_fit = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(I, d, k, args.cpu_users)
        if fit is not None and "error" not in fit:
            try:
                cpu_fit = cpu_baseline_fit(U, I, d, shard=FIT_SHARD_USERS, small_shard_seconds=oracle_small_shard_s,
                                           shard_seconds=oracle_shard_s)
            except Exception as exc:
                cpu_fit = {"error": repr(exc)}

    line = {
        "metric": "user-item predictions/sec", "value": value, "unit": "predictions/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "prewarm_steps": prewarm_steps + (2 if args.prewarm_seconds > 0 else 0),
        "ms_per_step": ms_per_step, "step_ms_by_hip_events": {"first": step_ms[0], "last": step_ms[-1], "min": min(step_ms),
                                                              "max": max(step_ms), "all": step_ms},
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        # arithmetic type of the dominant (MFMA) kernel
        # (the cascade computes in all three: int8 MFMA over every pair, bf16 MFMA over the ~4% it cannot rule out, fp32
        # chains over the survivors; what comes out is fp32-exact: "result_precision" and the "parity" block)
        "dtype": "fp32" if args.precision == "fp32" else ("i8/bf16/fp32" if cascade else "bf16"),
        "result_precision": ("fp32-exact (int8 MFMA pre-filter -> bf16 MFMA filter, both with proven error bounds -> fp32 "
                             "re-scoring of the survivors)" if cascade else
                             "fp32-exact (bf16 MFMA filter with a proven error bound + fp32 re-scoring of the survivors)")
                            if exact else args.precision,
        "data": "synthetic",
        "config": {"workload": "synthetic %d users x %d items, identity features, d=%d, LinearRepresentation + "
                               "DotProduct, biased, fused top-%d (BASELINE.json configs[2])" % (U, I, d, k),
                   "users": U, "items": I, "n_components": d, "top_k": k,
                   "parallelism": "items sharded x%d, users replicated" % world,
                   "exchange": None if world == 1 else ("all-to-all by user (floor + lists)" if topk_fn is sharding.sharded_top_k_a2a
                                                        else "all-gather (floor + lists)"),
                   "collective_selfcheck": selfcheck,
                   "topk_method": method,
                   "score_kernel_variant": "global_load_lds" if args.variant & 1 else "register-staged"},
        "roofline": roofline, "roofline_bf16_stage": roofline_bf16_stage, "roofline_k1": roofline_k1, "roofline_k1_multi_nnz": roofline_k1_multi, "cpu_baseline": cpu, "parity": parity,
        "fp32_mfma_mode": fp32_mode, "bf16_filter_mode": bf16_mode, "roofline_bf16_dense": roofline_bf16_dense,
        "public_api_mode": public_api, "trained_weights_mode": trained, "parity_fit": parity_fit, "parity_fit_binned": parity_fit_binned, "parity_multi_nnz": parity_multi, "configs": configs, "fit": fit,
        "roofline_fit": (fit or {}).get("roofline_fit"), "cpu_baseline_fit": cpu_fit,
    }
    # ---- what ONE rank of an 8-GPU run does per step, emulated on one GPU (scripts/rank_sim.py, scripts/fit_rank_sim.py: every
    # kernel at its per-rank size, the shared floor / the exchange plan real, nothing on the wire) -- read from the committed
    # profiles of the same code; the multi-GPU numbers themselves are the driver's to measure
    scale_emulation = None
    try:
        import glob
        rs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_rank_sim_n8.json")))
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_fit_rank_sim_n8.json")))
        scale_emulation = {"note": "one-GPU emulations of ONE rank of an 8-GPU run (committed profiles, not measured in this run)"}
        if rs:
            r_ = json.load(open(rs[-1]))
            scale_emulation["predict"] = {"file": os.path.basename(rs[-1]), "per_rank_step_ms": r_.get("per_rank_step_ms_at_N8_emulated"),
                                          "ideal_ms": ms_per_step / 8.0 if world == 1 else None,
                                          "kernels_ms": r_.get("kernels_ms"),
                                          "exchange_bytes_received_per_rank": r_.get("exchange_bytes_received_per_rank")}
        if fs:
            f_ = json.load(open(fs[-1]))
            scale_emulation["fit"] = {"file": os.path.basename(fs[-1]), "per_rank_compute_ms_per_step": f_.get("per_rank_compute_ms_per_step"),
                                      "plan": f_.get("plan"), "wire_bytes_per_rank_and_step": f_.get("wire_bytes_per_rank_and_step_total"),
                                      "round3_wire_bytes_per_rank_and_step": f_.get("round3_wire_bytes_per_rank_and_step")}
    except Exception as exc:
        scale_emulation = {"error": repr(exc)}
    line["scale_emulation"] = scale_emulation
    # dispatches per step the way rocprofv3 counts them (kernels + fills + copies): read from the committed kernel-stats of this
    # command when one exists for the current round; the HIP-event count (our own launches only) is always present
    line["launches_per_step_by_hip_events"] = len(events) / float(args.steps)
    # ---- the FULL record goes to a side file; stdout gets ONE compact line (bench_line.py: < 4 KB, scalars only) ----
    import bench_line
    full_path = None
    try:
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        full_path = os.path.join("gpurun_out", "bench_full.json" if world == 1 else "bench_full_n%d.json" % world)
        with open(os.path.join(ROOT, full_path), "w") as fh:
            json.dump(line, fh)
    except OSError:
        full_path = None
    sys.stdout.flush()
    print(bench_line.compact_line(line, full_path))
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
