#!/bin/bash
# last call of the round: the whole GPU suite (log kept), the fitted-weights diagnostics with second-call times
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
( DEBUG=0 EPOCH_BLOCKS=4,15,30 LR=0.1 OUT=gpurun_out/r03_diag_trained_lr01 timeout 600 python scripts/diag_trained.py > $OUT/diag_lr01.log 2>&1 ); echo "diag lr0.1 rc=$?"
( DEBUG=0 EPOCH_BLOCKS=20,60 LR=0.01 OUT=gpurun_out/r03_diag_trained_lr001 timeout 600 python scripts/diag_trained.py > $OUT/diag_lr001.log 2>&1 ); echo "diag lr0.01 rc=$?"
rm -f $OUT/r03_diag_trained_*.npz
