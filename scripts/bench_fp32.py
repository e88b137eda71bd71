import sys, os, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tensorrec_amd as T
from tensorrec_amd import ops
U, I, d, k = 65536, 1_000_000, 128, 10
g = torch.Generator(device="cuda"); g.manual_seed(0)
u = torch.randn((U, d), device="cuda", generator=g); v = torch.randn((I, d), device="cuda", generator=g)
ub = torch.randn(U, device="cuda", generator=g); ib = torch.randn(I, device="cuda", generator=g)
u_op, _, kpad = ops.score_prep(u, ops.DTYPE_F32); v_op, _, _ = ops.score_prep(v, ops.DTYPE_F32)
res = {}
for name, knob in (("pipelined", 1), ("generic", 0)):
    T._native.set_tuning("blockmax_pipelined_f32", knob)
    for _ in range(2): out = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_F32, kpad, k, ub, ib)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = ops.score_topk_two_stage(u_op, v_op, ops.DTYPE_F32, kpad, k, ub, ib)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    res[name] = out
    print(name, "%.1f ms  %.1f TFLOP/s (whole two-stage top-k)" % (dt * 1e3, 2.0 * U * I * d / dt / 1e12), flush=True)
assert torch.equal(res["pipelined"][0], res["generic"][0]) and torch.equal(res["pipelined"][1], res["generic"][1])
print("identical results")
