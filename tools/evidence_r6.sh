#!/bin/bash
# Round-6 evidence on the GPU box.  Outputs under gpurun_out/ev6/.
#   profiles : PMC traffic FIRST (so that the kernel-stats runs' own bench lines carry `traffic`), then rocprofv3 kernel stats of the default bench (config 2, f32) and of config 3 (bf16); PMC traffic of both (bench.py reads
#              roofline.traffic and roofline.hbm_bound's GB per step from the result)
#   paths    : the data-parallel code path with one forced RCCL rank and with two gloo ranks sharing the GPU (exposed-communication
#              and last-bucket fields), the Winograd fuzz, the codebook-search soak
# usage: tools/evidence_r5.sh [profiles|paths]   (no argument: both; one gpurun call holds at most 20 minutes)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ev6
mkdir -p $O
WHAT=${1:-all}
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" != paths ]; then
echo "== pmc traffic $(date +%T)"; bash $R/tools/pmc_traffic.sh 2>&1 | tee $O/pmc_traffic.log || { echo FAILED traffic; tail -5 $O/pmc_traffic.log; exit 1; }
echo "== pmc traffic, config 3 bf16 $(date +%T)"; bash $R/tools/pmc_traffic.sh c3 --config hifimusic --precision bf16 2>&1 | tee $O/pmc_traffic_c3.log || { echo FAILED traffic c3; exit 1; }
python3 - <<PY
import json
a = json.load(open("$R/gpurun_out/pmc_traffic/traffic.json"))
b = json.load(open("$R/gpurun_out/pmc_traffic_c3/traffic.json"))
assert a["_build"] == b["_build"]
wb = b.pop("_workload"); b.pop("_build")
a["_passes"] = {wb: b}
a["_note"] = "per-launch averages; top level: the pass on _workload (the default bench); _passes: further passes by workload key. bench.py quotes bytes only for a run of the same workload on the same kernel sources (_build)"
json.dump(a, open("$O/pmc_traffic.json", "w"), indent=1, sort_keys=True)
import shutil, os
os.makedirs("$R/profiles", exist_ok=True)
shutil.copy("$O/pmc_traffic.json", "$R/profiles/r6_pmc_traffic.json")      # the kernel-stats passes below read it: their lines carry traffic
PY
echo "== kernel stats, config 2 f32 $(date +%T)"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vq --no-alt --no-configs > $O/c2_under_rocprof.json 2> $O/c2.err || { echo FAILED c2; tail -3 $O/c2.err; exit 1; }
echo "== kernel stats, config 3 bf16 $(date +%T)"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/bench.py --config hifimusic --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-vq --no-alt > $O/c3_under_rocprof.json 2> $O/c3.err || { echo FAILED c3; tail -3 $O/c3.err; exit 1; }
fi
[ "$WHAT" = profiles ] && { echo DONE profiles $(date +%T); exit 0; }
cd $R
echo "== one forced RCCL rank $(date +%T)"; timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --force-collectives --no-cpu-baseline --no-vq --no-alt --no-configs > $O/rccl_forced_1rank_bench.json 2> $O/rccl.err || { echo FAILED rccl; tail -5 $O/rccl.err; exit 1; }
echo "== two gloo ranks on this GPU $(date +%T)"; MQGAN_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --no-vq --no-alt --no-configs > $O/gloo_2rank_rehearsal_bench.json 2> $O/gloo2.err || { echo FAILED gloo2; tail -5 $O/gloo2.err; exit 1; }
echo "== winograd fuzz $(date +%T)"; timeout -k 10 300 python3 tools/fuzz_winograd.py 60 5 > $O/fuzz_winograd.txt 2>&1 || { echo FAILED fuzz; tail -5 $O/fuzz_winograd.txt; exit 1; }
echo "== codebook-search soak $(date +%T)"; timeout -k 10 500 python3 tools/vq_soak.py 300 7 > $O/vq_soak.txt 2>&1 || { echo FAILED soak; tail -5 $O/vq_soak.txt; exit 1; }
echo "== layers $(date +%T)"; timeout -k 10 200 python3 tools/layer_profile.py > $O/layers_c2.txt 2>&1 || { echo FAILED layers; exit 1; }
echo DONE $(date +%T)
