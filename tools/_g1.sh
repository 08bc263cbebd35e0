set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_exchange.py -x -q > gpurun_out/r3/t_exchange.log 2>&1; echo "exchange rc=$?" 
tail -5 gpurun_out/r3/t_exchange.log
timeout -k 10 420 python tools/gloo_async_probe.py --ranks 4 --out gpurun_out/r3/hang --watch 40 > gpurun_out/r3/probe.log 2>&1; echo "probe rc=$?"
grep "^\[probe\]" gpurun_out/r3/probe.log
timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --force-collectives --no-alt --no-configs --no-vq --no-cpu-baseline > gpurun_out/r3/b_forced.json 2> gpurun_out/r3/b_forced.err; echo "forced rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-alt --no-configs --no-vq --no-cpu-baseline > gpurun_out/r3/b_plain.json 2> gpurun_out/r3/b_plain.err; echo "plain rc=$?"
python - <<'PY'
import json
for n in ("forced","plain"):
    try:
        l=json.loads(open(f"gpurun_out/r3/b_{n}.json").read().strip().splitlines()[-1])
        print(n, l["ms_per_step"], l["host_ms_per_step"], l.get("collectives"))
    except Exception as e: print(n, "ERR", e)
PY
