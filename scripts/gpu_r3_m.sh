#!/bin/bash
# round 3: the multi-rank bench functionally on one GPU (two gloo ranks, items sharded, fit data-parallel) + one rank of an 8-GPU predict, emulated
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 300 python scripts/rank_sim.py 2>&1 | tail -1 | cut -c1-1500
TREC_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --users 65536 --items 600000 --no-cpu-baseline --parity-users 64 2> $OUT/bench_2rank_gloo.err | grep "^{" > $OUT/bench_2rank_gloo.json
tail -3 $OUT/bench_2rank_gloo.err
python -c "
import json; d=json.load(open('$OUT/bench_2rank_gloo.json')); print('2-rank gloo bench:', d['ms_per_step'], d['parity'], d['config']['exchange'], (d['fit'] or {}).get('fit_epochs_per_sec'), (d['fit'] or {}).get('error'))"
