#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
for T in "blockmax_bf16_rdlate=0" "blockmax_bf16_rdlate=1" "blockmax_bf16_rdlate=0" "blockmax_bf16_rdlate=1"; do
( timeout 600 python bench.py --configs headline --no-fit --no-cpu-baseline --no-k1-multi --parity-users 64 --steps 5 --warmup 2 --tune $T > $OUT/bench_n.json 2> $OUT/bench_n.err ); tail -1 $OUT/bench_n.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_n.json').read().strip().splitlines()[-1])
o=d['roofline']['other_kernels_avg_ms']
print('$T', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), 'grouped', round(o['score_gemm_blockmax_grouped'],2), 'hot', round(o['score_gemm_blockmax_hot'],2), 'dense bf16', round(d['bf16_filter_mode']['stage1_avg_launch_ms'],2), d['parity']['topk_ids_bit_exact_vs_oracle'], d['bf16_filter_mode']['equals_timed_cascade_output'], d['fp32_mfma_mode']['equals_timed_exact_mode_output'])
PY
done
timeout 1200 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py tests/test_gpu_fuzz_kinds.py -q 2>&1 | tail -3
