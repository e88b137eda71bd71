#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40
