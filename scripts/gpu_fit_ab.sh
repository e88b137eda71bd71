#!/bin/bash
# A/B of library builds on the fit leg (1M x 1M, d = 128, WMRB): gpu_fit_ab.sh libA.so libB.so ...   (an argument with '=' is a
# tuning for the shipped library instead: gpu_fit_ab.sh group_pairs_staged=1 group_pairs_staged=0)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
for L in "$@"; do
case "$L" in *=*) ( timeout 600 python scripts/fit_only.py 3 $L > $OUT/fit_ab.json 2> $OUT/fit_ab.err );; *) ( TREC_HIP_LIB=$PWD/tensorrec_amd/$L timeout 600 python scripts/fit_only.py 3 > $OUT/fit_ab.json 2> $OUT/fit_ab.err );; esac
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/fit_ab.json') if l.startswith('{')][-1])
r=d['roofline_fit']
print('$L', 'epochs/s', round(d['fit_epochs_per_sec'],2), 'ms/epoch', round(1e3*d['sec_per_epoch'],2), 'fused ms', round(r['avg_launch_ms'],2), 'frac', round(r['frac'],3), r.get('other_kernels_avg_ms'))
PY
done
