#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py -x -q 2>&1 | tail -15
( timeout 600 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_c.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds_c.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/fuzz_kinds_at_scale.json'))
for r in d:
    c=r['cascade']; dbg=c['debug']
    print(r['kind'], 'casc %.1f bf16 %.1f x%.2f'%(c['ms'],r['bf16_filter']['ms'],r['cascade_over_gauss']), c['stats'].get('prefilter'), 'flagged',c['stats'].get('flagged_users'),
          'wanted %.3f rowmax %.3f hot %s'%(dbg.get('int8_pairs_wanted',0)/max(1,dbg.get('int8_pairs_total',1)), dbg.get('row_count_max',0)/r['users'], dbg.get('hot_superblocks')),
          [c[k] for k in c if k.startswith('equals')], [r['bf16_filter'][k] for k in r['bf16_filter'] if k.startswith('equals')])
PY
( timeout 600 python bench.py --no-fit --no-cpu-baseline --no-k1-multi --steps 3 --warmup 1 > $OUT/bench_c.json 2> $OUT/bench_c.err ); echo "bench rc=$?"; tail -3 $OUT/bench_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac']); print(d['roofline']['other_kernels_avg_ms']); print(d['parity']); print(d['roofline_bf16_stage'])
PY
