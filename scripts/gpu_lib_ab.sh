#!/bin/bash
# A/B of two builds of the library on the headline step, same box: gpu_lib_ab.sh libA.so libB.so ...
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
ARGS="--prewarm-seconds 0 --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 5 --warmup 2"
for L in "$@"; do
( TREC_HIP_LIB=$PWD/tensorrec_amd/$L timeout 600 python bench.py $ARGS $EXTRA_TUNE > $OUT/lib_ab.json 2> $OUT/lib_ab.err )
python - <<PY
import json
d=json.load(open('gpurun_out/bench_full.json'))      # (bench.py prints a compact line; the full record is in the side file)
o=d['roofline']['other_kernels_avg_ms']
print('$L', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), {k.replace('score_gemm_','').replace('topk_',''): round(v,2) for k,v in o.items()}, d['parity']['topk_ids_bit_exact_vs_oracle'])
PY
done
