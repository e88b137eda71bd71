#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py -x -q 2>&1 | tail -5
( KINDS=gauss,popular_bias,heavy_tail OUT=gpurun_out/fuzz_kinds_b.json timeout 600 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_b.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds_b.log
( EPOCH_BLOCKS=4,15,30 LR=0.1 OUT=gpurun_out/diag_b_lr01 timeout 900 python scripts/diag_trained.py > $OUT/diag_b_lr01.log 2>&1 ); echo "diag rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/diag_b_lr01.json'))
for st in d['stages']:
    for r in st['runs']:
        print(st['epochs'], r['mode'], round(r['ms'],1), r['stats'].get('prefilter'), 'flagged', r['stats'].get('flagged_users'), 'hot', r['debug'].get('hot_superblocks'), 'wanted', r['debug'].get('int8_pairs_wanted'))
PY
rm -f $OUT/diag_b_lr01*.npz
