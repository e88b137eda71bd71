#!/bin/bash
# Clock / power of the GPU while the bench's timed steps run (stage 1 of the top-k = 93% of a step): evidence for the
# power-limit reading of the MFMA roofline fraction.  Output: gpurun_out/r01_power_trace.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r01_power_trace.txt; mkdir -p gpurun_out
python bench.py --no-fit --no-cpu-baseline --steps 40 --warmup 2 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
echo "# rocm-smi samples every 0.5 s while: python bench.py --no-fit --no-cpu-baseline --steps 40 --warmup 2" > $OUT
for i in $(seq 1 60); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$(date +%s.%N | cut -c1-14)" >> $OUT
  rocm-smi -d 0 --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use|busy" >> $OUT
  sleep 0.5
done
wait $BP
python -c "import json; b=json.load(open('gpurun_out/power_bench.json')); print('# bench: ms_per_step', b['ms_per_step'], 'stage-1 frac', b['roofline']['frac'])" >> $OUT
tail -40 $OUT
