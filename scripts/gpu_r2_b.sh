#!/bin/bash
# round 2, call B: new tests (verbose) + kernel trace of the predict half of the bench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_filter.py tests/test_gpu_shapes.py tests/test_gpu_model.py} -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/new_tests.log 2>&1
echo "new tests rc=$?"; grep -v "^$" $OUT/new_tests.log | grep -v Warning | tail -${TAILN:-30}
REPO=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o r02 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit --parity-users 256 --no-fp32-mode > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
echo "rocprof rc=$?"; cat $OUT/prof_bench.json
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -30 $f; done
