#!/bin/bash
# round 2, call H: the sharded cascade (2 ranks on one GPU over gloo), then the default bench with the bf16-filter record
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dp_fit.py -q -x -k "top_k" -p no:cacheprovider 2>&1 | tail -5
TREC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --users 65536 --items 600000 --no-cpu-baseline --no-fit --parity-users 64 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2-rank gloo bench rc=$?"; cat $OUT/bench_2rank_gloo.json | cut -c1-2500; tail -3 $OUT/bench_2rank_gloo.err
timeout 900 python bench.py --no-fit --no-cpu-baseline --no-k1-multi > $OUT/bench_nofit.json 2> $OUT/bench_nofit.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench_nofit.json')); print(d['ms_per_step'], d['bf16_filter_mode'], d['roofline']['traffic'])"
