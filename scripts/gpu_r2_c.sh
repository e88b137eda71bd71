#!/bin/bash
# round 2, call C: whole GPU suite, 2-rank functional bench (gloo, one GPU), PMC passes, default bench
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 30 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
TREC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --users 65536 --items 300000 --no-cpu-baseline --no-fit --parity-users 64 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2-rank gloo bench rc=$?"; cat $OUT/bench_2rank_gloo.json | cut -c1-1500; tail -3 $OUT/bench_2rank_gloo.err
bash scripts/gpu_pmc_cmd.sh "scripts/k1_multi.py" r02_k1_multi_pmc_summary "spmm_csr" s3 s4 2>&1 | tail -6
bash scripts/gpu_pmc_cmd.sh "bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fit --parity-users 64 --no-fp32-mode --no-k1-multi" r02_pmc_summary "blockmax_pipe|score_gemm_kernel|filter_finish|select_blocks|collect_blocks|fill_groups|prep_filter|spmm_csr|spmm_one|seg_" s1 s2 s3 s4 2>&1 | tail -40
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
echo "bench default rc=$?"; cat $OUT/bench_full.json; tail -3 $OUT/bench_full.err
