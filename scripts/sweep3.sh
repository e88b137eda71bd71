# A/B of the BLOCKMAX kernels on the full workload: generic vs pipelined (3 / 2 workgroups per CU)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit $* 2>gpurun_out/sweep3.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '-> ms/step %.1f K2 %.1f TF (%.3f)' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']), d['parity'])" || tail -3 gpurun_out/sweep3.err; }
python -m pytest tests/test_gpu_kernels.py -x -q -k "two_stage or shard or topk" 2>&1 | tail -3
for a in "$@"; do run $a; done
