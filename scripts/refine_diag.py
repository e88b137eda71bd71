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
    out = (ctypes.c_uint64 * 8)()
    fn = lib.trec_refine_diag_read
    fn.argtypes, fn.restype = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int], ctypes.c_int
    assert fn(out, 1) == 0
    n = max(1, int(out[0]))
    us = lambda x: 0.01 * float(x) / n          # noqa: E731  (100 MHz wall clock -> microseconds per workgroup)
    print(json.dumps({"refine_workgroups": int(out[0]), "us_per_workgroup": {
        "prologue_to_operands_resident": us(out[1]), "tile_bodies": us(out[2]), "tile_waits_and_barriers": us(out[3]),
        "flush_and_maxima_stores": us(out[4]), "whole_life": us(out[5]),
        "hop_row_ids_and_first_tile (diag 16)": us(out[6]), "hop_user_rows (diag 16)": us(out[7])}}), file=sys.stderr)
