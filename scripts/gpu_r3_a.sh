#!/bin/bash
# Round 3, call A: what fitted weights / structured data do to the cascade (diagnostics), and the rocprof evidence for the
# dense bf16 filter kernel (north_star's score kernel): kernel stats + PMC passes of `bench.py --prefilter none`.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( EPOCH_BLOCKS=4,15,30 LR=0.1 OUT=gpurun_out/diag_trained_lr01 timeout 900 python scripts/diag_trained.py > $OUT/diag_trained_lr01.log 2>&1 ); echo "diag lr0.1 rc=$?"; tail -3 $OUT/diag_trained_lr01.log | cut -c1-600
( EPOCH_BLOCKS=20,60 LR=0.01 OUT=gpurun_out/diag_trained_lr001 timeout 900 python scripts/diag_trained.py > $OUT/diag_trained_lr001.log 2>&1 ); echo "diag lr0.01 rc=$?"; tail -3 $OUT/diag_trained_lr001.log | cut -c1-600
( timeout 900 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds_at_scale.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds_at_scale.log
BARGS="bench.py --prefilter none --no-fit --no-cpu-baseline --no-fp32-mode --no-k1-multi --steps 3 --warmup 1 --parity-users 64"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_bf16dense_prof -o r03 -- python $REPO/$BARGS > $OUT/r03_bf16dense_bench_under_rocprof.json 2> $OUT/r03_bf16dense_prof.err ); echo "rocprof rc=$?"
f=$(find $OUT/r03_bf16dense_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r03_bf16dense_kernel_stats.csv 2>/dev/null; head -8 $OUT/r03_bf16dense_kernel_stats.csv | cut -c1-220
bash scripts/gpu_pmc_cmd.sh "bench.py --prefilter none --no-fit --no-cpu-baseline --no-fp32-mode --no-k1-multi --steps 1 --warmup 0 --parity-users 64" r03_bf16dense_pmc_summary "blockmax_bf16x16|blockmax_pipe" s1 s3 s4 | tail -8 | cut -c1-400
rm -rf $OUT/r03_bf16dense_prof/*/*.db 2>/dev/null
du -sh $OUT | tail -1
