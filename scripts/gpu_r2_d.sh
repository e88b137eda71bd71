#!/bin/bash
# round 2, call D: new / changed tests with their prints kept, then the default bench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_filter.py tests/test_gpu_rank_large.py tests/test_gpu_shapes.py -m gpu -q --timeout 900 -p no:cacheprovider -s > $OUT/new_tests.log 2>&1
echo "new tests rc=$?"; grep -E "passed|failed|rank_rows|predict_rank_of|config|case " $OUT/new_tests.log | cut -c1-400
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
echo "bench default rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err
