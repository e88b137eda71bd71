#!/bin/bash
# same-box A/B of two builds of the library on the cascade probe: scripts/probe/lib_old.so vs the in-tree build, twice each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for L in scripts/probe/lib_old.so ""; do
    TREC_HIP_LIB=${L:+$PWD/$L} timeout 300 python scripts/probe_cascade.py > /tmp/ab.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/probe_cascade.json"))["int8_cascade"]
k=d["kernels_ms"]
print("${L:-new}".ljust(28), "call %.2f ms | i8 %.2f grouped %.2f select %.2f compact %.2f collect %.2f lists %.2f finish %.2f prep8 %.2f" % (d["ms_per_call"], k["score_gemm_blockmax_i8"], k["score_gemm_blockmax_grouped"], k["topk_select_blocks"], k["topk_rows_compact"], k["topk_collect_blocks"], k["score_gemm_topk_grouped"], k["topk_filter_finish"], k["score_prep_i8"]))
PY
  done
done
