#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py -x -q 2>&1 | tail -4
for T in "" "cascade_rcap_pct=50,cascade_max_refined_pct=45,cascade_max_hot=4000" "cascade_rcap_pct=35,cascade_max_refined_pct=45,cascade_max_hot=4000"; do
  ( TUNE=$T CHECK=0 KINDS=gauss,clustered256_03,scaled_rows,popular_bias OUT=gpurun_out/fuzz_kinds_e.json timeout 600 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_e.log 2>&1 ); echo "kinds [$T] rc=$?"; tail -1 $OUT/fuzz_kinds_e.log
done
for T in "cascade_rcap_pct=20" "cascade_rcap_pct=50" "cascade_rcap_pct=35"; do
  ( timeout 600 python bench.py --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 4 --warmup 1 --tune $T > $OUT/bench_e.json 2> $OUT/bench_e.err ); 
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_e.json').read().strip().splitlines()[-1])
print('$T', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), {k: round(v,2) for k,v in d['roofline']['other_kernels_avg_ms'].items()})
PY
done
