#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> bench (small, then BASELINE size, both staging variants) -> rocprof.
# Everything lands in gpurun_out/ (merged back by gpurun).  Each stage has its own timeout so a hang cannot eat the box.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out
STAGES="${1:-smoke tests bench prof}"
echo "== $(date) stages: $STAGES" > $OUT/summary.log
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/summary.log
nproc >> $OUT/summary.log

for st in $STAGES; do
case $st in
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
  echo "smoke rc=$?" >> $OUT/summary.log; tail -3 $OUT/smoke.log >> $OUT/summary.log ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 30 --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/summary.log; tail -40 $OUT/pytest_gpu.log >> $OUT/summary.log ;;
bench)
  for v in ${VARIANTS:-1 0 3 2 5 4 7 6}; do
    timeout 600 python bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline --no-fit > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err
    echo "bench full v$v rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/bench_v$v.json')); print('ms/step %.1f  K2 %.1f TF (%.3f)  K1 %.0f GB/s  parity %s' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_k1']['achieved'], d['parity']))" 2>&1)" >> $OUT/summary.log
    tail -2 $OUT/bench_v$v.err >> $OUT/summary.log
  done ;;
benchfull)
  timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
  echo "bench default rc=$?" >> $OUT/summary.log; cat $OUT/bench_full.json >> $OUT/summary.log; tail -3 $OUT/bench_full.err >> $OUT/summary.log
  timeout 600 python bench.py --steps 2 --warmup 1 --users 65536 --items 1000000 --precision fp32 --no-cpu-baseline > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
  echo "bench fp32 rc=$?" >> $OUT/summary.log; cat $OUT/bench_fp32.json >> $OUT/summary.log; tail -3 $OUT/bench_fp32.err >> $OUT/summary.log ;;
prof)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o r01 -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err )
  echo "rocprof rc=$?" >> $OUT/summary.log
  find $OUT/prof -name "*kernel_stats*" | head -3 >> $OUT/summary.log
  for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -12 $f >> $OUT/summary.log; done ;;
esac
done
echo "== done $(date)" >> $OUT/summary.log
cat $OUT/summary.log
