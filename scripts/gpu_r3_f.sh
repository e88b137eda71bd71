#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_cascade.py tests/test_gpu_filter.py -q 2>&1 | tail -40
