cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fit --parity-users 64 --no-fp32-mode --no-k1-multi $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.2f' % d['ms_per_step'], 'stage1 %.2f ms' % d['roofline']['avg_launch_ms'], 'frac %.4f' % d['roofline']['frac'], d['parity']['topk_ids_bit_exact_vs_oracle'])"; }
for i in 1 2; do
  TREC_HIP_LIB=$PWD/tensorrec_amd/libtensorrec_hip_oldstage1.so run "old kernel      "
  run "new, 2 buffers  " "--tune blockmax_nbuf=2"
  run "new, 3 buffers  " "--tune blockmax_nbuf=3"
done
