cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in ${SWEEP:-1 9 11}; do
  timeout 300 python bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline --no-fit ${BENCH_EXTRA} 2>gpurun_out/sweep_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v: ms/step %.1f K2 %.1f TF (%.3f)' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']))" || tail -3 gpurun_out/sweep_$v.err
done
