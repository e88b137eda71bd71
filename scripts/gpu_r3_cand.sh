#!/bin/bash
# candidate-list tail: diagnostics of the refining launch (tuning cascade_cand_diag; needs a library built with
# FLAGS_score_blockmax="-fno-honor-nans -DTREC_CAND_DIAG") and a kernel trace of one configuration
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_candidates.py tests/test_gpu_cascade.py -x -q 2>&1 | tail -4
bash scripts/gpu_ab.sh cascade_candidates=1 cascade_cand_diag=1 cascade_cand_diag=2 cascade_candidates=0
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/$OUT/prof_cand
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_cand -o cand -- python $GRAFT_REPO_ROOT/bench.py --configs headline --no-fit --no-cpu-baseline --no-k1-multi --no-fp32-mode --parity-users 64 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/prof_cand.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_cand/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last int8 launch starts the last step: list every kernel from there to the end with its duration and the gap before it
idx = [i for i, r in enumerate(rows) if 'blockmax_i8x16' in r['Kernel_Name']]
i0 = idx[-1]
prev_end = int(rows[i0]['Start_Timestamp'])
tot = 0
for r in rows[i0:i0 + 80]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.3f ms  gap %8.3f  %s' % ((e - s) / 1e6, (s - prev_end) / 1e6, r['Kernel_Name'][:100]))
    prev_end = max(prev_end, e)
PY
rm -rf gpurun_out/prof_cand/*/*.db 2>/dev/null; du -sh gpurun_out/prof_cand
