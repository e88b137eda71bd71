#!/bin/bash
# PMC passes over an arbitrary command (each pass its own rocprofv3 run: counters + --kernel-trace only).
# usage: gpu_pmc_cmd.sh "<python args>" <summary name> "<kernel-name substrings, |-separated>" [sets...]
#   results: gpurun_out/<summary name>.txt  (per kernel: per-dispatch averages of every counter)
# FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size
# (MI355X_MICROARCH.md, HBM section) -- the doubling is applied where the number is USED (bench.py), not here.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
CMD="$1"; NAME="${2:-pmc_cmd}"; FILTER="${3:-.}"; shift 3
OUT=$REPO/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
declare -A SETS
SETS[s1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH"
SETS[s2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SETS[s3]="FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
SETS[s4]="WRITE_SIZE TCC_HIT TCC_MISS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
for s in ${@:-s3 s4}; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $OUT/$s -o pmc -- python $REPO/$CMD > $OUT/$s.out 2> $OUT/$s.err )
  echo "$s rc=$?"
done
python - <<PY
import csv, glob, collections, re
out=open("$REPO/gpurun_out/$NAME.txt","w")
out.write("# rocprofv3 --pmc <set> --kernel-trace -- python $CMD ; per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KB as reported\n")
import sys, json; sys.path.insert(0, "$REPO"); import bench; out.write("# csrc_sha: %s\n" % json.dumps(bench.csrc_stamp(), sort_keys=True))
pat=re.compile(r"$FILTER")
for s in ("s1","s2","s3","s4"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%s, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add((k,r["Dispatch_Id"]))
        for k,_ in seen: cnt[k]+=1
        for k in agg:
            if pat.search(k):
                out.write("%s | dispatches=%d | "%(k,cnt[k])+" ".join("%s=%.5g"%(c,v/cnt[k]) for c,v in sorted(agg[k].items()))+"\n")
out.close()
print(open("$REPO/gpurun_out/$NAME.txt").read())
PY
