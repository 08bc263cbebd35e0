#!/bin/bash
# HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate passes) for `bench.py`; writes profiles-style JSON
# usage: bash tools/pmc_traffic.sh [tag [extra bench.py args]]    e.g.  pmc_traffic.sh c3 --config hifimusic --precision bf16
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-}; [ $# -gt 0 ] && shift
OUT=$R/gpurun_out/pmc_traffic${TAG:+_$TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  echo "pmc_traffic: pass $C $(date +%T)" | tee -a $OUT/progress.log
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vq --no-alt --no-configs "$@" > $OUT/$C.log 2>&1 || { echo "pass $C failed"; tail -5 $OUT/$C.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
import bench
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if "mq::" not in k: continue
    name = k.split("(")[0].replace("void ", "").replace("mq::", "")
    f, w = cs.get("FETCH_SIZE", [0]), cs.get("WRITE_SIZE", [0])
    # gfx950: FETCH_SIZE (KB) reports half of a wide coalesced read stream -> x2 (MI355X_MICROARCH.md, HBM section)
    out[name] = {"launches": len(f), "fetch_bytes_per_launch": 2 * 1024 * sum(f) / len(f), "write_bytes_per_launch": 1024 * sum(w) / len(w)}
out["_build"] = bench.kernel_build_id()
# the workload the pass ran (bench.py only quotes these bytes for a run of the same workload): from the bench line in the pass's log
for line in open("$OUT/FETCH_SIZE.log"):
    if line.startswith('{"metric"'):
        out["_workload"] = json.loads(line)["config"]["workload_key"]
assert "_workload" in out, "no bench line in $OUT/FETCH_SIZE.log"
json.dump(out, open("$OUT/traffic.json", "w"), indent=1, sort_keys=True)
out.pop("_build")
print("workload:", out.pop("_workload"))
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:12]:
    print(f"{k[:50]:50s} n={v['launches']:4d} fetch={v['fetch_bytes_per_launch']/1e6:9.1f} MB write={v['write_bytes_per_launch']/1e6:9.1f} MB")
PY
