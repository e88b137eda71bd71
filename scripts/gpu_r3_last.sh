#!/bin/bash
# the headline line and its rocprofv3 kernel stats with the round's final code (the budget's last two minutes)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03f_prof -o r03f -- python $REPO/bench.py --configs headline --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 256 > $OUT/r03_final_bench_under_rocprof.json 2> $OUT/r03f_prof.err ); echo "rocprof rc=$?"
f=$(find $OUT/r03f_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r03_final_kernel_stats.csv 2>/dev/null; rm -rf $OUT/r03f_prof
timeout 60 python bench.py --configs headline --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 256 > $OUT/r03_final_bench_headline.json 2> /dev/null; echo "bench rc=$?"
python - <<'PY'
import json
for f in ('r03_final_bench_under_rocprof','r03_final_bench_headline'):
    try:
        d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['roofline']['frac'], d['parity']['topk_ids_bit_exact_vs_oracle'])
    except Exception as e: print(f, 'failed', e)
PY
