#!/bin/bash
# round 4, first GPU pass: the tests the user-side rewrite touches, a short headline bench, a kernel trace for the dispatch count
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_user_prep.py tests/test_gpu_cascade.py tests/test_gpu_candidates.py tests/test_gpu_filter.py tests/test_gpu_fuzz_kinds.py tests/test_gpu_rccl_world1.py -x -q -m gpu > $OUT/r4a_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/r4a_pytest.log
timeout 600 python bench.py --configs headline --no-cpu-baseline --no-fit --steps 10 --warmup 3 > $OUT/r4a_bench.json 2> $OUT/r4a_bench.err; echo "bench rc=$?"; tail -3 $OUT/r4a_bench.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/r4a_bench.json"))
    print({k:d.get(k) for k in ("ms_per_step","value","step_ms_by_hip_events")})
    print("roofline", {k:d["roofline"].get(k) for k in ("frac","avg_launch_ms","other_kernels_avg_ms")})
    print("parity", d.get("parity"))
    print("public", d.get("public_api_mode"))
    print("checks", d["config"].get("checks"))
except Exception as e:
    print("parse failed", e); print(open("$OUT/r4a_bench.json").read()[-2000:])
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4a_prof -o r4a -- python $REPO/bench.py --configs headline --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 64 --steps 5 --warmup 2 > $OUT/r4a_bench_under_rocprof.json 2> $OUT/r4a_prof.err ); echo "rocprof rc=$?"
f=$(find $OUT/r4a_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/r4a_kernel_stats.csv 2>/dev/null; head -30 $OUT/r4a_kernel_stats.csv | cut -c1-150
t=$(find $OUT/r4a_prof -name "*kernel_trace.csv" | head -1); cp $t $OUT/r4a_kernel_trace.csv 2>/dev/null
