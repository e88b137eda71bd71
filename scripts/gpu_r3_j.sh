#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
for T in "i8_user_classes=1" "i8_user_classes=0" "i8_user_classes=1"; do
( timeout 600 python bench.py --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 6 --warmup 2 --tune $T > $OUT/bench_j.json 2> $OUT/bench_j.err ); tail -1 $OUT/bench_j.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_j.json').read().strip().splitlines()[-1])
o=d['roofline']['other_kernels_avg_ms']; k1=d['roofline_k1']['avg_launch_ms']
tot=sum(o.values())+d['roofline']['avg_launch_ms']
print('$T', 'ms_per_step', round(d['ms_per_step'],2), 'i8', round(d['roofline']['avg_launch_ms'],2), 'timed kernels sum', round(tot,2), {k: round(v,2) for k,v in o.items()}, d['parity']['filter'].get('refined_rows'))
PY
done
