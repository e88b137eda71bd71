"""RCCL on ONE GPU: every collective of the scoring / training path through the real "nccl" backend (= RCCL on ROCm) at
world size 1 -- as much of RCCL as a one-GPU box can execute (VERDICT r2 "missing" #5: until now RCCL itself had never run,
the multi-rank tests use gloo).  sharding.FORCE_COLLECTIVES makes the one-rank world take the multi-rank code paths: the
known-answer self-check, both exchange forms of the shared floor and of the list merge, an item-sharded predict_top_k through
the public API (int8 cascade with its floor / statistics / overflow reductions) and a data-parallel fit step with its gradient
all-reduce.  Runs in a spawned process (a process group per test process would leak into the other GPU tests)."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import tensorrec_amd as T
        from tensorrec_amd import sharding, ops
        sharding.FORCE_COLLECTIVES = True
        out = {"backend": dist.get_backend(), "selfcheck": sharding.collective_selfcheck(dev)}
        out["a2a_available"] = bool(sharding.a2a_available(torch.zeros(1, device=dev)))
        # both forms of the two per-query exchanges agree with each other and with the local answer
        g = torch.Generator(device=dev); g.manual_seed(0)
        sel_max = torch.randn((10, 5000), device=dev, generator=g)
        f_ag = sharding.shared_topk_floor(sel_max)
        f_a2a = sharding.shared_topk_floor_a2a(sel_max)
        out["floor_forms_equal"] = bool(torch.equal(f_ag, f_a2a) and torch.equal(f_ag, sel_max.min(dim=0).values))
        vals = torch.randn((5000, 10), device=dev, generator=g).sort(dim=1, descending=True).values
        idx = torch.argsort(torch.rand((5000, 1000), device=dev, generator=g), dim=1)[:, :10].to(torch.int32)
        v1, i1 = sharding.sharded_top_k(vals, idx, 10)
        v2, i2 = sharding.sharded_top_k_a2a(vals, idx, 10, replicate=True)
        out["topk_forms_equal"] = bool(torch.equal(v1, v2) and torch.equal(i1, i2) and torch.equal(v1, vals))
        # item-sharded predict_top_k through the public API: 300,000 items, d = 64 -> the int8 cascade with every reduction
        rng = np.random.RandomState(1)
        n_u, n_i, d = 150, 300_000, 64
        uf = sp.random(n_u, 30, density=0.2, random_state=rng, format="csr", dtype=np.float32)
        itf = sp.identity(n_i, format="csr", dtype=np.float32)
        model = T.TensorRec(n_components=d, seed=11)
        model.build(uf.shape[1], itf.shape[1])
        w = model.get_weights()
        w["item_feature_biases"] = (0.05 * rng.standard_normal(w["item_feature_biases"].shape)).astype(np.float32)
        model.set_weights(w)
        sv, si = model.predict_top_k(uf, itf, k=10, item_sharded=True, item_offset=0)
        out["sharded_stage1"] = ops.LAST_FILTER_STATS.get("prefilter")
        sharding.FORCE_COLLECTIVES = False
        pv, pi = model.predict_top_k(uf, itf, k=10)
        sharding.FORCE_COLLECTIVES = True
        out["sharded_predict_equals_plain"] = bool(np.array_equal(sv, pv) and np.array_equal(si, pi))
        # one data-parallel WMRB step (gradient all-reduce over RCCL) == the plain step
        inter = sp.random(n_u, 2000, density=0.02, random_state=rng, format="csr", dtype=np.float32)
        inter.data[:] = 1.0
        itf2 = sp.identity(2000, format="csr", dtype=np.float32)

        def fit(dp):
            m = T.TensorRec(n_components=16, loss_graph=T.loss_graphs.WMRBLossGraph(), seed=3, data_parallel=dp, deterministic=True)
            m.fit(inter, uf, itf2, epochs=2, learning_rate=0.05, n_sampled_items=20)
            return m.get_weights()
        w_dp, w_plain = fit(True), fit(False)
        # (same sums in the same order except that the DP step keeps loss and L2 gradients apart until after the all-reduce)
        out["dp_fit_equals_plain"] = bool(all(np.allclose(w_dp[k], w_plain[k], rtol=1e-4, atol=1e-6) for k in w_plain))
        # ... and with every exchange form of the gradient plan on RCCL: a ReLU tower's dense layers are untagged -> with the
        # threshold lowered they are reduce-scattered (reduce_scatter_tensor), stepped by row range and all-gathered IN PLACE
        # (all_gather_into_tensor); the tables multiplied with features are "disjoint" (no exchange; broadcast at the end)
        sharding.SHARD_MIN_NUMEL = 64

        def fit_relu(dp):
            m = T.TensorRec(n_components=16, user_repr_graph=T.representation_graphs.ReLURepresentationGraph(),
                            loss_graph=T.loss_graphs.WMRBLossGraph(), seed=3, data_parallel=dp, deterministic=True)
            m.fit(inter, uf, itf2, epochs=2, learning_rate=0.05, n_sampled_items=20)
            modes = dict(m._dp_plan.mode) if dp else None
            if dp:
                m.dp_sync(optimizer_state=True)
            return m.get_weights(), modes
        (w_dp, modes), (w_plain, _) = fit_relu(True), fit_relu(False)
        out["dp_plan_modes"] = sorted(set(modes.values()))
        out["dp_relu_fit_equals_plain"] = bool(all(np.allclose(w_dp[k], w_plain[k], rtol=1e-4, atol=1e-6) for k in w_plain))
        ret.update(out)
    finally:
        dist.destroy_process_group()


def test_every_collective_runs_through_rccl_at_world_size_one():
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), ret), nprocs=1, join=True)
    ret = dict(ret)
    assert ret["backend"] == "nccl"
    assert ret["selfcheck"] == "ok" and ret["a2a_available"], ret
    assert ret["floor_forms_equal"] and ret["topk_forms_equal"], ret
    assert ret["sharded_predict_equals_plain"], ret
    assert str(ret["sharded_stage1"]).startswith("int8"), ret
    assert ret["dp_fit_equals_plain"], ret
    assert "sharded" in ret["dp_plan_modes"] and "disjoint" in ret["dp_plan_modes"], ret
    assert ret["dp_relu_fit_equals_plain"], ret
