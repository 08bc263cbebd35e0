#!/bin/bash
# MFMA / VALU / LDS counters of the conv kernels on single layers (tools/h2_micro.py), separate --pmc passes.
# usage (on the GPU box): bash tools/pmc_h2.sh <tag> <script in tools/: h2_micro.py | f32_micro.py> <args...>     -> gpurun_out/pmc_<tag>.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
SCRIPT=${1:-h2_micro.py}; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  echo "pmc_$TAG: pass $i $(date +%T)" | tee -a $OUT/progress.log
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/$SCRIPT "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, cs in agg.items():
    if "conv" not in k and "l2_" not in k: continue
    name = k.split("(")[0].replace("void ", "").replace("mq::", "")
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches": len(next(iter(cs.values()))), "avg_ns_under_pmc": (sum(dur[k]) / len(dur[k])) if dur.get(k) else None}
    e.update({c: a[c] for c in sorted(a)})
    if a.get("SQ_INSTS_MFMA"): e["valu_per_mfma"] = a.get("SQ_INSTS_VALU", 0) / a["SQ_INSTS_MFMA"]
    if a.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_frac"] = a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"]
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"): e["mfma_busy_per_sq_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]
    if "FETCH_SIZE" in a: e["fetch_bytes_per_launch_x2"] = 2 * 1024 * a["FETCH_SIZE"]      # gfx950: FETCH_SIZE counts half (MI355X_MICROARCH.md)
    if "WRITE_SIZE" in a: e["write_bytes_per_launch"] = 1024 * a["WRITE_SIZE"]
    out[name] = e
json.dump(out, open("$R/gpurun_out/pmc_$TAG.json", "w"), indent=1, sort_keys=True)
for k, e in out.items():
    print(k, {x: (round(y, 4) if isinstance(y, float) else y) for x, y in e.items() if x in ("valu_per_mfma", "lds_conflict_frac", "mfma_busy_per_sq_busy", "launches", "fetch_bytes_per_launch_x2", "write_bytes_per_launch")})
PY
