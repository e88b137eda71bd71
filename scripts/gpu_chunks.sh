#!/bin/bash
# the int8 launch's item chunks per user block (grid = user blocks x chunks): step time per value
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
for C in "$@"; do
( timeout 600 python bench.py --prewarm-seconds 0 --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 5 --warmup 2 --chunks $C > $OUT/bench_ab.json 2> $OUT/bench_ab.err ); tail -1 $OUT/bench_ab.err | grep -v amdgpu.ids
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_ab.json').read().strip().splitlines()[-1])
o=d['roofline']['other_kernels_avg_ms']
print('chunks $C', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), 'select', round(o.get('topk_select_blocks',0),2), d['parity']['topk_ids_bit_exact_vs_oracle'])
PY
done
