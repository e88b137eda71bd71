#!/bin/bash
# Diagnostics build of the refining launch (make -C tensorrec_amd/csrc diag): per-workgroup clock sums and the A/B switches of
# cascade_cand_diag (1 queues dropped, 2 atomics without stores, +4 first item tile requested before the user gathers, +8 no
# maxima stores).  usage: gpu_refine_diag.sh 0 4 8 1 ...
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
export TREC_HIP_LIB=$PWD/tensorrec_amd/libtensorrec_hip_diag.so
ARGS="--prewarm-seconds 0 --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 5 --warmup 2"
for V in "$@"; do
( timeout 600 python scripts/refine_diag.py $ARGS --tune cascade_cand_diag=$V $EXTRA_TUNE > $OUT/refine_diag_$V.json 2> $OUT/refine_diag_$V.err )
grep refine_workgroups $OUT/refine_diag_$V.err > $OUT/refine_diag_clk_$V.json
python - <<PY
import json
d=json.loads([l for l in open('$OUT/refine_diag_$V.json') if l.startswith('{')][-1])
o=d['roofline']['other_kernels_avg_ms']
print('cand_diag=$V', 'ms_per_step', round(d['ms_per_step'],2), {k.replace('score_gemm_','').replace('topk_',''): round(v,2) for k,v in o.items()}, d['parity']['topk_ids_bit_exact_vs_oracle'])
print(open('$OUT/refine_diag_clk_$V.json').read().strip())
PY
done
