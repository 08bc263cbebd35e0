#!/bin/bash
# usage: tools/pmc.sh <tag> <python script + args...>   (run on the GPU box; counters in separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not k.startswith("void mq::conv") and not k.startswith("mq::"): continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
