#!/bin/bash
# round 2, call A: new tests (verbose), whole GPU suite, default bench, PMC of the fit kernel
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_shapes.py -m gpu -q --timeout 600 -p no:cacheprovider -s > $OUT/new_tests.log 2>&1
echo "new tests rc=$?"; grep -v "^$" $OUT/new_tests.log | tail -40
timeout 1500 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 30 --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_shapes.py --deselect tests/test_gpu_filter.py > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
echo "bench default rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err
bash scripts/gpu_pmc_cmd.sh "scripts/fit_only.py 1" r02_fit_pmc_summary "wmrb_user_fused|seg_fill|spmm_csr|adam|sample_items" s3 s4 2>&1 | tail -30
