cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fit $* 2>gpurun_out/sweep2.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '-> ms/step %.1f K2 %.1f TF (%.3f)' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']), d['roofline'].get('other_kernels_avg_ms'), d['parity'])" || tail -3 gpurun_out/sweep2.err; }
for a in "$@"; do run $a; done
