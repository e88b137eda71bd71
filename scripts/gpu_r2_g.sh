#!/bin/bash
# round 2, call G: whole GPU suite with the cascade in the product path, smoke, the 8-rank emulation, then the evidence refresh
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 1800 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 30 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python scripts/rank_sim.py 2>&1 | tail -3
TREC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --users 65536 --items 600000 --no-cpu-baseline --no-fit --parity-users 64 2> $OUT/bench_2rank_gloo.err | grep "^{" > $OUT/bench_2rank_gloo.json
python -c "
import json; d=json.load(open('$OUT/bench_2rank_gloo.json')); print('2-rank gloo bench:', d['ms_per_step'], d['parity']['topk_ids_bit_exact_vs_oracle'], d['parity']['topk_values_bit_exact_vs_oracle'], d['parity']['filter'].get('prefilter'))"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o r02 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-users 256 --no-fp32-mode --no-k1-multi > $REPO/$OUT/prof_bench.json 2> $REPO/$OUT/prof.err )
echo "rocprof rc=$?"; for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -30 $f | cut -c1-200; done
bash scripts/gpu_pmc_cmd.sh "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fit --parity-users 64 --no-fp32-mode --no-k1-multi" r02_pmc_summary "blockmax_i8|blockmax_pipe|blockmax_bf16x16|rows_collect|rows_status|sb_scale|sb_reduce|bias_i8|user_err|sumsq|score_gemm_kernel|filter_finish|select_blocks|collect_blocks|fill_groups|prep_filter|prep_i8|rows_|spmm_csr|spmm_one|seg_" s1 s2 s3 s4 > $OUT/pmc_predict.log 2>&1
grep -c . $OUT/r02_pmc_summary.txt
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
echo "bench default rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err
