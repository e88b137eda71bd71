#!/bin/bash
# PMC passes (each its own rocprofv3 run, counters + --kernel-trace only) over a short bench invocation.
# usage: gpu_pmc.sh "<bench args>" ; results: gpurun_out/pmc/<set>/...counter_collection.csv + pmc_summary.txt
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
OUT=$REPO/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${1:---steps 1 --warmup 1 --users 262144 --items 262144 --no-cpu-baseline --no-fit}"
declare -A SETS
SETS[s1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH"
SETS[s2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SETS[s3]="FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
SETS[s4]="WRITE_SIZE TCC_HIT TCC_MISS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
for s in ${PMC_SETS:-s1 s2 s3 s4}; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $OUT/$s -o pmc -- python $REPO/bench.py $ARGS > $OUT/$s.json 2> $OUT/$s.err )
  echo "$s rc=$?"
done
python - <<PY
import csv, glob, collections
out=open("$OUT/../pmc_summary.txt","w")
for s in ("s1","s2","s3","s4"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%s, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            seen.add((k,r["Dispatch_Id"]))
        for k,_ in seen: cnt[k]+=1
        for k in agg:
            if "score_gemm" in k or "spmm_csr" in k or "topk_merge" in k or "score_prep" in k:
                out.write("%s | %s | dispatches=%d | "%(s,k,cnt[k])+" ".join("%s=%.4g"%(c,v/cnt[k]) for c,v in sorted(agg[k].items()))+"\n")
out.close()
print(open("$OUT/../pmc_summary.txt").read())
PY
