#!/bin/bash
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_profile.sh r03 2>&1 | tail -60
rm -rf gpurun_out/r03_prof/*/*.db gpurun_out/r03_pmc_*/*/*.db 2>/dev/null
du -sh gpurun_out
