#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py tests/test_gpu_fuzz_kinds.py tests/test_gpu_dp_fit.py tests/test_gpu_rccl_world1.py tests/test_gpu_rank_large.py -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( CHECK=0 timeout 600 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_i.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds_i.log
( timeout 600 python bench.py --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 256 --steps 5 --warmup 2 > $OUT/bench_i.json 2> $OUT/bench_i.err ); tail -2 $OUT/bench_i.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_i.json').read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), {k: round(v,2) for k,v in d['roofline']['other_kernels_avg_ms'].items()}, d['parity']['topk_ids_bit_exact_vs_oracle'], d['parity']['filter'])
PY
