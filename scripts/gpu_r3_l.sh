#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
true
for T in "blockmax_i8_rdlate=0" "blockmax_i8_rdlate=1" "blockmax_i8_rdlate=2" "blockmax_i8_rdlate=3" "blockmax_i8_rdlate=1" "blockmax_i8_rdlate=2" "blockmax_i8_rdlate=3"; do
( timeout 600 python bench.py --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 6 --warmup 2 --tune $T > $OUT/bench_l.json 2> $OUT/bench_l.err ); tail -1 $OUT/bench_l.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_l.json').read().strip().splitlines()[-1])
o=d['roofline']['other_kernels_avg_ms']
print('$T', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), 'grouped', round(o['score_gemm_blockmax_grouped'],2), 'hot', round(o['score_gemm_blockmax_hot'],2), d['parity']['topk_ids_bit_exact_vs_oracle'], d['parity']['topk_values_bit_exact_vs_oracle'])
PY
done
