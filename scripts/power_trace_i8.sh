#!/bin/bash
# Clock / power while ONLY the int8 stage-0 kernel (then only the bf16 stage-1 kernel) runs back to back for ~8 s.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r02_power_trace_i8.txt; mkdir -p gpurun_out; : > $OUT
for K in i8 bf16; do
  LOOPS=$([ $K = i8 ] && echo 90 || echo 50) KERNEL=$K python scripts/i8_loop.py > gpurun_out/loop_$K.txt 2>/dev/null &
  BP=$!
  echo "# rocm-smi samples every 0.5 s while the $K kernel runs back to back" >> $OUT
  for i in $(seq 1 80); do
    kill -0 $BP 2>/dev/null || break
    rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT; echo >> $OUT
    sleep 0.5
  done
  wait $BP
  cat gpurun_out/loop_$K.txt >> $OUT
done
grep -E "1[0-9]{3}Mhz|2[0-9]{3}Mhz|ms per launch|^#" $OUT | tail -60
