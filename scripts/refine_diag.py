"""Diagnostics build only (csrc built with -DTREC_CAND_DIAG, TREC_HIP_LIB pointing at it): run bench.py's headline step in
this process, then print where the workgroups of the refining launch spent their lives (g_refine_clk, score_blockmax.hip).
usage: TREC_HIP_LIB=.../libtensorrec_hip_diag.so python scripts/refine_diag.py <bench.py arguments>"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                      # noqa: E402

if __name__ == "__main__":
    bench.main()
    from tensorrec_amd import _native as N
    lib = N.load()
    import numpy as np
    n_wg = 1 << 17
    out = (ctypes.c_uint64 * (n_wg * 4))()
    fn = lib.trec_refine_diag_read
    fn.argtypes, fn.restype = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_longlong, ctypes.c_int], ctypes.c_int
    assert fn(out, n_wg * 4, 0) == 0
    a = np.frombuffer(out, dtype=np.uint64).reshape(n_wg, 4).astype(np.float64)
    a = a[a[:, 3] > 0] * 0.01            # workgroups of the last refining launch; 100 MHz wall clock -> microseconds
    print(json.dumps({"refine_workgroups_last_launch": int(a.shape[0]), "us_per_workgroup_mean": {
        "prologue_to_operands_resident": float(a[:, 0].mean()), "tile_loop": float(a[:, 1].mean()),
        "superblock_end_flush_and_maxima": float(a[:, 2].mean()), "whole_life": float(a[:, 3].mean())},
        "whole_life_percentiles_10_50_90": [float(x) for x in np.percentile(a[:, 3], [10, 50, 90])]}), file=sys.stderr)
