#!/bin/bash
# A/B of tuning knobs on the headline step: gpu_ab.sh "knob=0" "knob=1" ...   (TESTS=1 first runs the cascade / filter / fuzz tests)
# The FIRST bench process on a fresh box runs ~4 ms per step slower than the following ones whatever it measures: a throw-away
# run goes first (WARM=0 skips it), and the order of the arguments should alternate when the difference is small.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
if [ "$TESTS" = "1" ]; then timeout 1500 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py tests/test_gpu_fuzz_kinds.py tests/test_gpu_dp_fit.py tests/test_gpu_candidates.py -q 2>&1 | tail -4; fi
ARGS="--prewarm-seconds 0 --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 5 --warmup 2"
if [ "$WARM" != "0" ]; then timeout 600 python bench.py $ARGS > /dev/null 2>&1; fi
for T in "$@"; do
( timeout 600 python bench.py $ARGS --tune $T > $OUT/bench_ab.json 2> $OUT/bench_ab.err ); tail -1 $OUT/bench_ab.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.load(open('gpurun_out/bench_full.json'))     # (bench.py prints a compact line; the full record is in the side file)
o=d['roofline']['other_kernels_avg_ms']
i8=d['roofline']['avg_launch_ms']*d['roofline'].get('launches_per_step',1)
print('$T', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(i8,2), 'others', round(sum(o.values()),2), 'rest', round(d['ms_per_step']-i8-sum(o.values()),2), {k.replace('score_gemm_','').replace('topk_',''): round(v,2) for k,v in o.items()}, d['parity']['topk_ids_bit_exact_vs_oracle'], d['parity']['topk_values_bit_exact_vs_oracle'], d['parity']['filter'].get('flagged_users'))
PY
done
