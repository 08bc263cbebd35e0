#!/bin/bash
# Round-3 evidence on the GPU box: rocprofv3 kernel stats of the default bench (config 2, f32) and of config 3 (bf16), PMC
# traffic of both (bench.py reads roofline.traffic from the result), MFMA / LDS / traffic counters of the codebook-search
# kernels.  Outputs under gpurun_out/ev3/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats, config 2 f32 $(date +%T)"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vq --no-alt --no-configs > $O/c2_under_rocprof.json 2> $O/c2.err || { echo FAILED c2; tail -3 $O/c2.err; exit 1; }
echo "== kernel stats, config 3 bf16 $(date +%T)"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/bench.py --config hifimusic --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-vq --no-alt > $O/c3_under_rocprof.json 2> $O/c3.err || { echo FAILED c3; tail -3 $O/c3.err; exit 1; }
echo "== pmc traffic $(date +%T)"; bash $R/tools/pmc_traffic.sh 2>&1 | tee $O/pmc_traffic.log || { echo FAILED traffic; tail -5 $O/pmc_traffic.log; exit 1; }
echo "== pmc traffic, config 3 bf16 $(date +%T)"; bash $R/tools/pmc_traffic.sh c3 --config hifimusic --precision bf16 2>&1 | tee $O/pmc_traffic_c3.log || { echo FAILED traffic c3; exit 1; }
python3 - <<PY
import json
a = json.load(open("$R/gpurun_out/pmc_traffic/traffic.json"))
b = json.load(open("$R/gpurun_out/pmc_traffic_c3/traffic.json"))
assert a["_build"] == b["_build"]
for k, v in b.items():
    if not k.startswith("_") and k not in a:
        a[k] = dict(v, run="config 3 (hifimusic, bf16)")
a["_note"] = "per-launch averages; kernels without 'run' were measured in the config-2 f32 bench, those with 'run' in the named pass"
json.dump(a, open("$O/pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
echo "== counters, codebook search $(date +%T)"; bash $R/tools/pmc_h2.sh vq vq_micro.py 2>&1 | tee $O/counters_vq.log || { echo FAILED counters; tail -5 $O/counters_vq.log; exit 1; }
cp $R/gpurun_out/pmc_vq.json $O/
echo DONE $(date +%T)
