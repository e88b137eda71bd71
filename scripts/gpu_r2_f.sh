#!/bin/bash
# round 2, call F: whole GPU suite, then the committed evidence of the final code (kernel trace, PMC passes, default bench)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; OUT=gpurun_out; export TMPDIR=/tmp; REPO=$PWD
timeout 1800 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 30 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_r2_e.sh
