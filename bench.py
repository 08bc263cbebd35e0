#!/usr/bin/env python3
"""Headline benchmark: one optimisation step of the PreEncoder VQ-GAN (generator forward,
discriminator step, generator step, clip + Adam for both) on synthetic 80-mel x 512-frame batches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line from rank 0.  `value` = mel-frames/s over all N GPUs (weak scaling: batch 32 per
GPU), inputs resident in HBM before the timed region.  Extra objects:
  roofline      - the dominant kernel (an implicit-GEMM conv instantiation) timed live with HIP events
                  on its launch stream: algorithmic FLOPs / launch time vs the dense f32 MFMA peak
  vq            - the quantiser kernels the metric names: FSQ (HBM-bound streaming) and the generic
                  L2 nearest-neighbour search (f32-MFMA-bound), each against its own roof
  cpu_baseline  - the CPU oracle's train step (kind "port") on a bounded sample, rank 0 / N = 1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "mel-frames/sec train step (80×512, codebook 1024) at 1/2/4/8 GPU; VQ argmin GB/s"
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def load_config(name):
    with open(os.path.join(ROOT, "configs", f"model_config_{name}.yaml")) as f:
        return yaml.safe_load(f)


def synthetic_batches(B, T, mel, n, device, rank):
    from mqgan_amd.train import synthetic_mel
    out = []
    for i in range(n):
        x = np.stack([synthetic_mel(T, mel, rank * 100000 + i * B + b) for b in range(B)])
        out.append(torch.from_numpy(x).to(device))
    return out


def pmc_traffic(kernel_tag):
    """HBM-side bytes per launch of `kernel_tag` from the last committed PMC pass (tools/pmc_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate rocprofv3 --pmc runs of this same bench, FETCH_SIZE x2 as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the process, so the figure is taken
    from profiles/; None when no pass covers this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    e = t.get(kernel_tag)
    if not e:
        return None, None
    return e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"], os.path.relpath(files[-1], ROOT)


def time_events(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_vq(device):
    """FSQ at N = 2^24 vectors (HBM roof) and L2-argmin at the BASELINE codebook shapes (f32 MFMA roof)."""
    from mqgan_amd import ops
    from mqgan_amd.quantizer import FSQ
    out = {}
    q = FSQ([8, 5, 5, 5]).to(device)
    n = 1 << 24
    g = torch.Generator(device="cpu").manual_seed(7)
    z = (torch.randn(n, 4, generator=g) * 2.0).to(device)
    with torch.no_grad():
        q(z)
        ms = time_events(lambda: q(z), 20)
    gb = 36.0 * n / 1e9                     # 16 B z in + 16 B codes + 4 B index per vector
    out["fsq"] = {"n": n, "ms": ms, "achieved_GBps": gb / (ms / 1e3), "peak_GBps": PEAK_HBM_GBS,
                  "frac": gb / (ms / 1e3) / PEAK_HBM_GBS, "bound": "hbm", "bytes_per_vector": 36}
    del z
    rows = 16384
    for k, d in ((1024, 256), (2048, 512), (8192, 512)):
        g = torch.Generator(device="cpu").manual_seed(7)
        x = torch.randn(rows, d, generator=g).to(device)
        g = torch.Generator(device="cpu").manual_seed(11)
        cb = torch.randn(k, d, generator=g).to(device)
        cn = ops.l2_codebook_norms(cb)
        ws = ops._ws_bytes(ops.L().mq_l2_argmin_workspace(rows), x)
        ops.l2_argmin(x, cb, cn, workspace=ws)
        ms = time_events(lambda: ops.l2_argmin(x, cb, cn, workspace=ws), 10)
        flops = 2.0 * rows * k * d
        byts = 4.0 * (rows * d + k * d + rows)
        out[f"l2_argmin_{k}x{d}"] = {"n": rows, "ms": ms, "achieved_TFLOPs": flops / (ms / 1e3) / 1e12,
                                     "peak_TFLOPs": PEAK_F32_MFMA_TFLOPS, "frac": flops / (ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                     "bound": "mfma", "algorithmic_GBps": byts / (ms / 1e3) / 1e9}
    return out


def cpu_baseline(seconds_budget=30.0):
    """The oracle's train step on the host cores: hifispeech architecture, mel 80, batch 2 x 256 frames
    (BASELINE config 1), discriminators active, generator in train mode.  One untimed tiny step to page
    things in, then as many timed steps as fit the budget (at least one)."""
    from oracle import train_step as TS
    from mqgan_amd.train import synthetic_mel
    cfg = load_config("hifispeech")
    threads = torch.get_num_threads()
    Pg = TS.make_leaves(TS.init_generator_params(cfg))
    pc, mc = cfg["model"]["discriminator_patch"], cfg["model"]["discriminator_multibin"]
    Pp = TS.make_leaves(TS.init_patch_params(80, pc["hidden_channels"], pc["kernel_sizes"]))
    Pm = TS.make_leaves(TS.init_multibin_params(80, mc["n_bins"], mc["hidden_channels"], mc["kernel_sizes"]))
    tr = TS.OracleTrainer(Pg, Pp, Pm, cfg, gen_training=True)
    warm = torch.from_numpy(np.stack([synthetic_mel(32, 80, 900 + b) for b in range(1)]))
    tr.step(warm, torch.tensor([32], dtype=torch.int32))
    B, T = 2, 256
    n, t0 = 0, time.time()
    while True:
        x = torch.from_numpy(np.stack([synthetic_mel(T, 80, n * B + b) for b in range(B)]))
        tr.step(x, torch.tensor([T] * B, dtype=torch.int32))
        n += 1
        el = time.time() - t0
        if el + el / n > seconds_budget:
            break
    return {"value": n * B * T / el, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} step(s) of batch {B} x {T} frames, hifispeech@80, D active, fp32, oracle/train_step.py, "
                      f"{el:.1f} s on {threads} threads ({os.cpu_count()} logical CPUs visible)"}


def quick_engine(cfg, precision, device, batches, lens, B, T, steps=4, warmup=1):
    """Step time of the same workload on another conv engine (training.precision)."""
    import copy
    from mqgan_amd.train import Trainer
    c = copy.deepcopy(cfg)
    c["training"]["precision"] = precision
    tr = Trainer(c, device=device, build_data=False)
    tr.generator.train(); tr.patch_discriminator.train(); tr.multibin_discriminator.train()
    tr.epoch = 1
    for i in range(warmup):
        tr.train_step(batches[i % len(batches)], lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = tr.train_step(batches[i % len(batches)], lens)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": B * T * steps / dt, "unit": "mel-frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "losses_finite": all(np.isfinite(float(v)) for v in last.values()),
           "arithmetic": {"fp32x3": "f32 operands split exactly into 3 bf16 terms in-kernel, 6 cross products on the bf16 MFMA, "
                                    "f32 accumulate; passes the reference fixtures at the f32 tolerances",
                          "bf16": "operands rounded to bf16 in-kernel, f32 accumulate, f32 tensors (reference autocast mode)"}[precision]}
    del tr
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="hifispeech", choices=["hifispeech", "hifimusic"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--frames", type=int, default=None, help="frames per item (default: the config's crop_len)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp32x3"],
                    help="matrix engine of the convs (fp32 = the BASELINE metric's; bf16 = BASELINE config 3, secondary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vq", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary fp32x3 / bf16 engine timings")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    args = ap.parse_args()

    import __graft_entry__ as ge
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if rank == 0:
        ge.build()
    # MQGAN_BENCH_SHARE_GPU=1 (rehearsal only): all ranks on device 0 with the gloo backend, to exercise the N > 1
    # code path on a one-GPU box; RCCL needs one device per rank.
    share = os.environ.get("MQGAN_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        dist.barrier()
    from mqgan_amd import _lib, ops
    from mqgan_amd.train import Trainer
    _lib.load()                                   # fails loudly if the HIP library is missing

    cfg = load_config(args.config)
    B = args.batch or cfg["data"]["batch_size"]
    T = args.frames or cfg["data"]["crop_len"]
    mel = cfg["model"]["mel_channels"]
    cfg["training"]["precision"] = args.precision
    tr = Trainer(cfg, device=device, build_data=False)
    tr.generator.train()
    tr.patch_discriminator.train()
    tr.multibin_discriminator.train()
    tr.epoch = 1
    batches = synthetic_batches(B, T, mel, 4, device, rank)
    lens = torch.full((B,), T, dtype=torch.int32, device=device)

    for i in range(args.warmup):
        tr.train_step(batches[i % len(batches)], lens)
    timer = ops.KernelTimer()
    ops.TIMER = timer
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = tr.train_step(batches[i % len(batches)], lens)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    finite = all(np.isfinite(float(v)) for v in last.values())

    if rank == 0:
        frames = world * B * T * args.steps
        summ = timer.summary()
        total_conv_ms = sum(v["ms"] for v in summ.values())
        dom_tag, dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        ach = dom["flops"] / (dom["ms"] / 1e3) / 1e12
        traffic, traffic_src = pmc_traffic(dom_tag)
        # the split engine spends 6 bf16 MFMAs per f32 product block: its ceiling is a sixth of the bf16 peak
        peak = PEAK_BF16_MFMA_TFLOPS if "bf16" in dom_tag else (PEAK_BF16_MFMA_TFLOPS / 6 if "_x3_" in dom_tag else PEAK_F32_MFMA_TFLOPS)
        roof = {"bound": "mfma", "kernel": dom_tag, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_flop_per_launch": dom["flops"] / dom["launches"],
                "launches_per_step": dom["launches"] / args.steps,
                "avg_launch_ms": dom["ms"] / dom["launches"], "share_of_step_time": dom["ms"] / 1e3 / elapsed,
                "all_conv_kernels": {k: {"ms_per_step": v["ms"] / args.steps, "TFLOPs": v["flops"] / (v["ms"] / 1e3) / 1e12,
                                         "launches_per_step": v["launches"] / args.steps} for k, v in sorted(summ.items())},
                "conv_share_of_step_time": total_conv_ms / 1e3 / elapsed}
        line = {"metric": METRIC, "value": frames / elapsed, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 convs (f32 accumulate, f32 tensors) / f32 VQ",
                                              "fp32x3": "f32 via 3-term bf16 split (6 products, f32 accumulate) / f32 VQ"}[args.precision],
                "data": "synthetic",
                "config": {"workload": f"{args.config} PreEncoder VQ-GAN train step (G fwd + D step + G step + clip/Adam), "
                                       f"{mel} mels x {T} frames, batch {B}/GPU, FSQ {cfg['model']['generator']['fsq_levels']}, "
                                       "D active, FM off, train-mode dropout", "global_batch": world * B, "frames": T,
                           "parallelism": f"dp{world}"},
                "losses_finite": bool(finite), "roofline": roof}
        if world == 1 and not args.no_alt and args.precision == "fp32":
            # secondary figures, same workload, opt-in engines (never `value`): see DESIGN.md section 3
            del tr
            torch.cuda.empty_cache()
            line["other_engines"] = {p: quick_engine(cfg, p, device, batches, lens, B, T) for p in ("fp32x3", "bf16")}
        if not args.no_vq:
            line["vq"] = bench_vq(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
