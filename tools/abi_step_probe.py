"""Per-step wall time of the config-2 step with the collectives forced through the C ABI's RCCL entry points (comm.RcclAbiComm,
one rank) against the plain step, same process: the communicator is created first, then both trainers run."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = bench.load_config("hifispeech")
B, T = cfg["data"]["batch_size"], cfg["data"]["crop_len"]
bench.RUNNING_WORKLOAD = bench.workload_key("hifispeech", B, T, "fp32")
for label, kw in (("plain", {}), ("abi forced", {"force": True, "abi": True}), ("plain again", {})):
    tr, batches, lens, timer, elapsed, last = bench.timed_run(cfg, "fp32", B, T, 3, 2, dev, 1, 0, **kw)
    per = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_step(batches[i % len(batches)], lens)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        per.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    print(f"{label}: timed_run {elapsed / 3 * 1e3:.1f} ms/step; then per step (host issue ms, wall ms): " +
          ", ".join(f"({a:.0f}, {b:.0f})" for a, b in per), flush=True)
    if kw:
        tr.comm.close()
    del tr, batches
    torch.cuda.empty_cache()
