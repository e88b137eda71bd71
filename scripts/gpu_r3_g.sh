#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
( timeout 1500 python bench.py > $OUT/bench_g.json 2> $OUT/bench_g.err ); echo "bench rc=$?"; tail -5 $OUT/bench_g.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_g.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
for k in ('parity','trained_weights_mode','parity_fit','parity_multi_nnz','roofline_bf16_dense','cpu_baseline_fit'):
    print(k, json.dumps(d.get(k))[:1500])
for k,v in (d.get('configs') or {}).items(): print(k, json.dumps(v)[:1500])
PY
