"""K8 (dense Adam, 28 bytes per element) on one 1M x 128 table: variants of the kernel's memory policy (tuning adam_variant: bit 0
non-temporal stores, bit 1 non-temporal loads, bit 2 two float4 per array in flight) x grid caps (adam_blocks).  HIP events, 20 launches."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorrec_amd import _native as N

if __name__ == "__main__":
    n = 1_000_000 * 128
    g_ = torch.Generator(device="cuda"); g_.manual_seed(0)
    w0 = torch.randn(n, device="cuda", generator=g_); grad = 0.01 * torch.randn(n, device="cuda", generator=g_)
    ref = None
    for blocks in (4096, 16384, 1 << 20):
        for variant in (0, 1, 2, 3, 4, 5, 7):
            N.set_tuning("adam_variant", variant); N.set_tuning("adam_blocks", blocks)
            w, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
            N.call("trec_adam_tf_step", N.ptr(w), N.ptr(m), N.ptr(v), N.ptr(grad), n, 0.1, 0.9, 0.999, 1e-8, 1e-5)
            if ref is None:
                ref = (w.clone(), m.clone(), v.clone())
            same = bool(torch.equal(w, ref[0]) and torch.equal(m, ref[1]) and torch.equal(v, ref[2]))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                N.call("trec_adam_tf_step", N.ptr(w), N.ptr(m), N.ptr(v), N.ptr(grad), n, 0.1, 0.9, 0.999, 1e-8, 1e-5)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 20
            print(json.dumps({"adam_blocks": blocks, "adam_variant": variant, "ms": round(ms, 4), "TB_per_s": round(28.0 * n / ms / 1e9, 3),
                              "bit_identical_to_variant_0": same}), flush=True)
