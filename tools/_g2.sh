set -x
mkdir -p gpurun_out/r3/stuck
export MQGAN_BENCH_SHARE_GPU=1 MQGAN_BENCH_TRACE=1 MQGAN_DDP_TRACE=1 MQGAN_DDP_ASYNC=1 MQGAN_BENCH_TRACE_AFTER=200 MQGAN_BENCH_TRACE_DIR=gpurun_out/r3/stuck
timeout -k 10 330 python bench.py --gpus 4 --steps 4 --warmup 2 --no-alt --no-configs --no-vq --no-cpu-baseline > gpurun_out/r3/dp4_async.json 2> gpurun_out/r3/dp4_async.err; echo "dp4 async rc=$?"
tail -3 gpurun_out/r3/dp4_async.err
ls gpurun_out/r3/stuck
