#!/bin/bash
# round 2, call E: refresh of the committed evidence -- kernel trace + PMC of the predict path, fit PMC, default bench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp; REPO=$PWD
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o r02 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-users 256 --no-fp32-mode --no-k1-multi > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
echo "rocprof rc=$?"; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -24 $f | cut -c1-200; done
bash scripts/gpu_pmc_cmd.sh "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fit --parity-users 64 --no-fp32-mode --no-k1-multi" r02_pmc_summary "blockmax_pipe|score_gemm_kernel|filter_finish|select_blocks|collect_blocks|fill_groups|prep_filter|spmm_csr|spmm_one|seg_" s1 s2 s3 s4 > $OUT/pmc_predict.log 2>&1
grep -c . $OUT/r02_pmc_summary.txt
bash scripts/gpu_pmc_cmd.sh "scripts/fit_only.py 1" r02_fit_pmc_summary "wmrb_user_fused|seg_fill|spmm_csr|adam|sample_items" s3 s4 > $OUT/pmc_fit.log 2>&1
grep "wmrb_user_fused" $OUT/r02_fit_pmc_summary.txt | cut -c1-300
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
echo "bench default rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err
