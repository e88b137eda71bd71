#!/bin/bash
# round 4, second GPU pass: the whole GPU test tier, the per-rank emulations of the 8-GPU predict and fit
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/r4b_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/r4b_pytest.log | cut -c1-250
timeout 300 python scripts/rank_sim.py > $OUT/r4b_rank_sim.log 2>&1; echo "rank_sim rc=$?"; tail -3 $OUT/r4b_rank_sim.log | cut -c1-1500
timeout 300 python scripts/fit_rank_sim.py > $OUT/r4b_fit_rank_sim.log 2>&1; echo "fit_rank_sim rc=$?"; tail -3 $OUT/r4b_fit_rank_sim.log | cut -c1-2500
