#!/bin/bash
# Round-end evidence: default bench line, rocprofv3 kernel stats of the same command, PMC passes for HBM traffic, the per-rank
# emulations of the 8-GPU predict and fit (scripts/rank_sim.py, scripts/fit_rank_sim.py), configs[4]'s fit under rocprofv3 + PMC.
# STAGES (default all but the last three): bench prof pmc sims cfg4 tests | ab rank2 diag     PMC_SETS=2 limits the PMC passes to the
# two HBM / L2 sets (FETCH_SIZE; WRITE_SIZE TCC_HIT TCC_MISS)
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-r01}
STAGES="${STAGES:-bench prof pmc sims cfg4 tests}"
has() { case " $STAGES " in *" $1 "*) return 0;; *) return 1;; esac; }
if has sims; then
timeout 300 python scripts/rank_sim.py > $OUT/${TAG}_rank_sim.log 2>&1; echo "rank_sim rc=$?"; cp $OUT/rank_sim.json $OUT/${TAG}_rank_sim_n8.json 2>/dev/null; tail -1 $OUT/${TAG}_rank_sim.log | cut -c1-900
timeout 300 python scripts/fit_rank_sim.py > $OUT/${TAG}_fit_rank_sim.log 2>&1; echo "fit_rank_sim rc=$?"; cp $OUT/fit_rank_sim.json $OUT/${TAG}_fit_rank_sim_n8.json 2>/dev/null; tail -1 $OUT/${TAG}_fit_rank_sim.log | cut -c1-900
mkdir -p profiles; cp $OUT/${TAG}_rank_sim_n8.json $OUT/${TAG}_fit_rank_sim_n8.json profiles/ 2>/dev/null      # (the bench line below reads them: scale_emulation)
fi
if has bench; then
timeout 1200 python bench.py ${BENCH_ARGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json | cut -c1-6000
cp $OUT/bench_full.json $OUT/${TAG}_bench_full.json 2>/dev/null      # (stdout is the compact driver line; this is the full record)
fi
if has cfg4; then
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg4_prof -o cfg4 -- python $REPO/scripts/profile_cfg4.py 3 > $OUT/${TAG}_cfg4.log 2> $OUT/${TAG}_cfg4.err ); echo "cfg4 rocprof rc=$?"; tail -1 $OUT/${TAG}_cfg4.log
f=$(find $OUT/${TAG}_cfg4_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_cfg4_fit_kernel_stats.csv 2>/dev/null; head -8 $OUT/${TAG}_cfg4_fit_kernel_stats.csv | cut -c1-160
for s in "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS"; do       # (all four in one pass exceed what the hardware collects at once)
  n=$(echo $s | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/${TAG}_cfg4_pmc_$n -o pmc -- python $REPO/scripts/profile_cfg4.py 1 > /dev/null 2> $OUT/${TAG}_cfg4_pmc_$n.err ); echo "cfg4 pmc $n rc=$?"
done
python - <<PY
import csv, glob, collections
out=open("$OUT/${TAG}_cfg4_pmc_summary.txt","w")
out.write("# rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT TCC_MISS --kernel-trace -- python scripts/profile_cfg4.py 1 ; per-dispatch averages (KB)\n")
import sys, json; sys.path.insert(0, "$REPO"); import bench; out.write("# csrc_sha: %s\n" % json.dumps(bench.csrc_stamp(), sort_keys=True))
for f in sorted(glob.glob("$OUT/${TAG}_cfg4_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add((k,r["Dispatch_Id"]))
    cnt=collections.Counter(k for k,_ in seen)
    for k in agg:
        if any(x in k for x in ("split","pair_score","spmm","wmrb","gemm","seg_","adam")):
            out.write("%s | dispatches=%d | "%(k,cnt[k])+" ".join("%s=%.5g"%(c,v/cnt[k]) for c,v in sorted(agg[k].items()))+"\n")
out.close(); print(open("$OUT/${TAG}_cfg4_pmc_summary.txt").read()[:3000])
PY
fi
if has ab; then          # item chunks of the int8 launch (workgroup rounds / tail): AB="13 26 52"
for c in ${AB:-13 26}; do
  timeout 300 python bench.py --configs headline --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 64 --steps 8 --warmup 2 --chunks $c > $OUT/${TAG}_ab_chunks$c.json 2> /dev/null
  python -c "import json;d=json.load(open('$OUT/${TAG}_ab_chunks$c.json'));print('chunks $c: ms/step %.2f  int8 %.2f ms  frac %.4f  parity %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['checks']['topk_ids_bit_exact']))"
done
fi
if has tests; then
timeout 1200 python -m pytest tests -q -m gpu > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/${TAG}_pytest_gpu.log
fi
if has prof; then
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o $TAG -- python $REPO/bench.py --configs headline --no-cpu-baseline --no-fit --prewarm-seconds 0 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err ); echo "rocprof rc=$?"
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_kernel_stats.csv 2>/dev/null; head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
fi
if has pmc; then
PMC_N=0
for s in "FETCH_SIZE" "WRITE_SIZE TCC_HIT TCC_MISS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  PMC_N=$((PMC_N + 1)); if [ -n "$PMC_SETS" ] && [ $PMC_N -gt $PMC_SETS ]; then break; fi
  n=$(echo $s | cut -d' ' -f1)
  ( cd /tmp && timeout 900 rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$n -o pmc -- python $REPO/bench.py --configs headline --steps 1 --warmup 0 --prewarm-seconds 0 --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 64 > /dev/null 2> $OUT/${TAG}_pmc_$n.err ); echo "pmc $n rc=$?"
done
python - <<PY
import csv, glob, collections
out=open("$OUT/${TAG}_pmc_summary.txt","w")
out.write("# rocprofv3 --pmc <set> --kernel-trace -- python bench.py --configs headline --steps 1 --warmup 0 --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 64 ; per-dispatch averages\n")
import sys, json; sys.path.insert(0, "$REPO"); import bench; out.write("# csrc_sha: %s\n" % json.dumps(bench.csrc_stamp(), sort_keys=True))
out.write("# FETCH_SIZE / WRITE_SIZE are in KB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md)\n")
for f in sorted(glob.glob("$OUT/${TAG}_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add((k,r["Dispatch_Id"]))
    cnt=collections.Counter(k for k,_ in seen)
    for k in agg:
        if any(x in k for x in ("score_gemm","blockmax","spmm","topk","select_blocks","collect_blocks","rows_","fill_groups","prep_","filter_","seg_","natscale","bias_i8","prerefine","finish","cascade_floor","dense_users")):
            out.write("%s | dispatches=%d | "%(k,cnt[k])+" ".join("%s=%.5g"%(c,v/cnt[k]) for c,v in sorted(agg[k].items()))+"\n")
out.close(); print(open("$OUT/${TAG}_pmc_summary.txt").read())
PY
fi
if has pmcx; then          # the counters bench.py quotes for the OTHER kernels: fit leg, dense bf16 filter kernel, K1 on 20 non-zeros per row
bash scripts/gpu_pmc_cmd.sh "scripts/fit_only.py 1" ${TAG}_fit_pmc_summary "wmrb|spmm|seg_|group|adam|sample_items|stage" s3 s4 > /dev/null; tail -n +3 $OUT/${TAG}_fit_pmc_summary.txt | cut -c1-260
bash scripts/gpu_pmc_cmd.sh "bench.py --configs headline --prefilter none --steps 1 --warmup 0 --prewarm-seconds 0 --no-cpu-baseline --no-fit --no-fp32-mode --no-k1-multi --parity-users 64" ${TAG}_bf16dense_pmc_summary "blockmax_bf16|select_blocks|collect|finish" s3 s4 > /dev/null; tail -n +3 $OUT/${TAG}_bf16dense_pmc_summary.txt | cut -c1-260
bash scripts/gpu_pmc_cmd.sh "scripts/k1_multi.py" ${TAG}_k1_multi_pmc_summary "spmm" s3 s4 > /dev/null; tail -n +3 $OUT/${TAG}_k1_multi_pmc_summary.txt | cut -c1-260
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fit_prof -o fit -- python $REPO/scripts/fit_only.py 4 > $OUT/${TAG}_fit_prof.log 2> $OUT/${TAG}_fit_prof.err ); echo "fit rocprof rc=$?"
f=$(find $OUT/${TAG}_fit_prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/${TAG}_fit_kernel_stats.csv 2>/dev/null; head -9 $OUT/${TAG}_fit_kernel_stats.csv | cut -c1-160
fi
if has rank2; then         # two ranks over gloo on ONE GPU, started BARE (bench.py launches its own ranks): the world > 1 path end to end (functional, not a scaling number)
TREC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --users 200000 --configs headline --no-cpu-baseline 2> $OUT/${TAG}_bench_2rank.err | grep -v "^\[Gloo\]" > $OUT/${TAG}_bench_2rank_gloo_one_gpu.json; echo "rank2 rc=$?"
fi
if has diag; then          # per-workgroup clock stamps of the refining launch (make -C tensorrec_amd/csrc diag first)
bash scripts/gpu_refine_diag.sh 0; cp $OUT/refine_diag_clk_0.json $OUT/${TAG}_refine_clocks.json 2>/dev/null
fi
