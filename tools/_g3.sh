set -x
mkdir -p gpurun_out/r3/probe
for n in 1 2 4; do
  timeout -k 10 400 python tools/gloo_async_probe.py --ranks $n --out gpurun_out/r3/probe/n$n --variants busy,busysync --watch 120 > gpurun_out/r3/probe/n$n.log 2>&1; echo "probe n=$n rc=$?"
  grep "^\[probe" gpurun_out/r3/probe/n$n.log
done
export MQGAN_BENCH_SHARE_GPU=1 MQGAN_BENCH_TRACE=1 MQGAN_DDP_TRACE=1 MQGAN_BENCH_TRACE_AFTER=200 MQGAN_BENCH_TRACE_DIR=gpurun_out/r3/stuck_fixed
timeout -k 10 330 python bench.py --gpus 4 --steps 4 --warmup 2 --no-alt --no-configs --no-vq --no-cpu-baseline > gpurun_out/r3/dp4_fixed.json 2> gpurun_out/r3/dp4_fixed.err; echo "dp4 fixed rc=$?"
unset MQGAN_BENCH_SHARE_GPU MQGAN_BENCH_TRACE MQGAN_DDP_TRACE
cat gpurun_out/r3/dp4_fixed.json | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['n_gpus'], l['ms_per_step'], l['host_ms_per_step'], l['collectives'])"
timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --force-collectives --no-alt --no-configs --no-vq --no-cpu-baseline > gpurun_out/r3/b_forced.json 2> gpurun_out/r3/b_forced.err; echo "forced rc=$?"
cat gpurun_out/r3/b_forced.json | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['n_gpus'], l['ms_per_step'], l['host_ms_per_step'], l['collectives'])"
