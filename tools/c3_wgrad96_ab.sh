#!/bin/bash
# config-3 A/B of the 96-channel weight-gradient kernel, alternating on one box
mkdir -p gpurun_out/s2
for i in 1 2; do for m in new old; do
  if [ $m = old ]; then export MQGAN_H2_NO_WGRAD96=1; else unset MQGAN_H2_NO_WGRAD96; fi
  timeout -k 10 200 python bench.py --config hifimusic --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-vq --no-alt 2>/dev/null > gpurun_out/s2/c3_$m$i.json || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/s2/c3_$m$i.json").read().strip().split("\n")[-1])
k = d["roofline"]["all_conv_kernels"]
print("$m", round(d["ms_per_step"], 2), {n: (round(v["ms_per_step"], 2), round(v["TFLOPs"])) for n, v in k.items() if "wgrad_h2" in n}, flush=True)
PY
done; done
