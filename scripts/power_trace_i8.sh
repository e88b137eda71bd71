#!/bin/bash
# Clock / power while ONLY one stage-0 / stage-1 kernel runs back to back for ~8 s:
#   int8 16x16x64 (192 and 128 users per wave), int8 32x32x32, bf16 32x32x16
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/r02_power_trace_i8.txt; mkdir -p gpurun_out; : > $OUT
for CFG in "i8 1 192 1" "i8 0 128 1" "bf16 1 128 33" "bf16 1 128 1"; do
  set -- $CFG
  LOOPS=$([ $1 = i8 ] && echo 90 || echo 45) KERNEL=$1 MFMA=$2 USERS=$3 VARIANT=$4 python scripts/i8_loop.py > gpurun_out/loop.txt 2>/dev/null &
  BP=$!
  echo "# rocm-smi samples every 0.5 s while: KERNEL=$1 MFMA=$2 (int8: 1 = 16x16x64, 0 = 32x32x32) USERS=$3 per wave VARIANT=$4 (bf16: 33 = 16x16x32, 1 = 32x32x16)" >> $OUT
  for i in $(seq 1 80); do
    kill -0 $BP 2>/dev/null || break
    rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed 's/=\+//g; s/GPU\[0\]\s*://g' >> $OUT; echo >> $OUT
    sleep 0.5
  done
  wait $BP
  cat gpurun_out/loop.txt >> $OUT
done
grep -E "1[0-9]{3}Mhz|2[0-9]{3}Mhz|ms per launch|^#" $OUT | awk 'NR%3==0 || /ms per|^#/' | tail -60
