#!/bin/bash
# Round-3 evidence in one call: whole GPU suite, smoke, fitted-weights diagnostics (clean times), every data kind at scale,
# the multi-rank bench on one GPU (gloo) + the 8-rank emulation, then gpu_profile.sh (bench line, rocprofv3 kernel stats, PMC
# passes) and the dense bf16 kernel's own stats + PMC passes.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( DEBUG=0 EPOCH_BLOCKS=4,15,30 LR=0.1 OUT=gpurun_out/r03_diag_trained_lr01 timeout 900 python scripts/diag_trained.py > $OUT/diag_lr01.log 2>&1 ); echo "diag lr0.1 rc=$?"
( DEBUG=0 EPOCH_BLOCKS=20,60 LR=0.01 OUT=gpurun_out/r03_diag_trained_lr001 timeout 900 python scripts/diag_trained.py > $OUT/diag_lr001.log 2>&1 ); echo "diag lr0.01 rc=$?"
rm -f $OUT/r03_diag_trained_*.npz
( OUT=gpurun_out/r03_fuzz_kinds_at_scale.json timeout 900 python scripts/fuzz_kinds_at_scale.py > $OUT/fuzz_kinds.log 2>&1 ); echo "kinds rc=$?"; tail -1 $OUT/fuzz_kinds.log
bash scripts/gpu_r3_m.sh 2>&1 | tail -4
cp $OUT/rank_sim.json $OUT/r03_rank_sim_n8.json; cp $OUT/bench_2rank_gloo.json $OUT/r03_bench_2rank_gloo_one_gpu.json
bash scripts/gpu_profile.sh r03 > $OUT/profile.log 2>&1; grep -E "rc=" $OUT/profile.log
if [ "$BF16DENSE" = "1" ]; then
BARGS="bench.py --configs headline --prefilter none --no-fit --no-cpu-baseline --no-fp32-mode --no-k1-multi --steps 3 --warmup 1 --prewarm-seconds 0 --parity-users 64"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_bf16dense_prof -o r03 -- python $REPO/$BARGS > $OUT/r03_bf16dense_bench_under_rocprof.json 2> $OUT/r03_bf16dense_prof.err ); echo "bf16 dense rocprof rc=$?"
f=$(find $OUT/r03_bf16dense_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r03_bf16dense_kernel_stats.csv 2>/dev/null
bash scripts/gpu_pmc_cmd.sh "bench.py --configs headline --prefilter none --no-fit --no-cpu-baseline --no-fp32-mode --no-k1-multi --steps 1 --warmup 0 --prewarm-seconds 0 --parity-users 64" r03_bf16dense_pmc_summary "blockmax_bf16x16|blockmax_pipe" s1 s3 s4 | tail -4 | cut -c1-300
fi
# the tail behind the int8 stage: candidate lists (default) against the table-driven tail, and the two-stream user batches
WARM=0 bash scripts/gpu_ab.sh cascade_candidates=1 cascade_candidates=0 cascade_candidates=1 cascade_candidates=0 cascade_user_batches=4 2>&1 | tee $OUT/r03_ab_tail.txt | cut -c1-60
find $OUT -name "*.db" -delete 2>/dev/null; du -sh $OUT | tail -1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'i8 ms', d['roofline']['avg_launch_ms'])
print('trained', d['trained_weights_mode']['ms_per_step'], d['trained_weights_mode']['stage1'], d['trained_weights_mode']['parity']['topk_ids_bit_exact_vs_oracle'], 'parity_fit', d['parity_fit']['green'], 'fit', d['fit']['fit_epochs_per_sec'])
PY
