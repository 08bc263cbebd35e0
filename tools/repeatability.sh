#!/bin/bash
# both configurations five times each on ONE box, alternating (final build): the same-box spread
mkdir -p gpurun_out/s3
for i in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vq --no-alt --no-configs 2>/dev/null > gpurun_out/s3/rep_c2_$i.json || exit 1
  timeout -k 10 200 python bench.py --config hifimusic --precision bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-vq --no-alt 2>/dev/null > gpurun_out/s3/rep_c3_$i.json || exit 1
  python - <<PY
import json
a = json.loads(open("gpurun_out/s3/rep_c2_$i.json").read().strip().split("\n")[-1])
b = json.loads(open("gpurun_out/s3/rep_c3_$i.json").read().strip().split("\n")[-1])
print("run $i: config 2 f32 %.2f ms/step (dominant %s %.3f)   config 3 bf16 %.2f ms/step (dominant %s %.3f)" % (
    a["ms_per_step"], a["roofline"]["kernel"], a["roofline"]["frac"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"]), flush=True)
PY
done
