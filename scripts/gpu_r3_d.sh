#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py -x -q 2>&1 | tail -15
( KINDS=gauss,scaled_rows,clustered256_03 OUT=gpurun_out/fuzz_kinds_d.json timeout 600 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_d.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds_d.log
( EPOCH_BLOCKS=4,15 LR=0.1 OUT=gpurun_out/diag_d_lr01 timeout 900 python scripts/diag_trained.py > $OUT/diag_d_lr01.log 2>&1 ); echo "diag rc=$?"; tail -3 $OUT/diag_d_lr01.log | cut -c1-300
( EPOCH_BLOCKS=20 LR=0.01 OUT=gpurun_out/diag_d_lr001 timeout 900 python scripts/diag_trained.py > $OUT/diag_d_lr001.log 2>&1 ); echo "diag rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/diag_d_lr01.json','gpurun_out/diag_d_lr001.json'):
    d=json.load(open(f))
    for st in d['stages']:
        for r in st['runs']:
            print(f[-12:], st['epochs'], r['mode'], round(r['ms'],1), r['stats'].get('prefilter'), 'flagged', r['stats'].get('flagged_users'), 'after wide', r['stats'].get('flagged_after_wide_pass'), 'hot', r['debug'].get('hot_superblocks'), 'eq', st['cascade_equals_bf16_filter'])
PY
rm -f $OUT/diag_d_*.npz
