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


def kernel_build_id():
    """Hash of the sources of the kernels whose traffic is quoted (convolutions, transforms, the pointwise family): ties a PMC traffic
    file to the build it was measured on.  Comments and blank lines do not count (a corrected remark must not orphan the counters):
    '//' to the end of the line outside string literals, '/* */' blocks."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in ("gemm_core.h", "igemm.hip", "igemm_h2.hip", "winogradn.hip", "plan.cpp", "wino1d.hip", "winow.hip", "wino_tables.h", "common.h", "pointwise.hip"):
        with open(os.path.join(ROOT, "mqgan_amd", "csrc", f), "r") as fh:
            text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
        for line in text.splitlines():
            line = re.sub(r'//(?=(?:[^"]*"[^"]*")*[^"]*$).*', "", line).rstrip()
            if line.strip():
                h.update(line.encode() + b"\n")
    return h.hexdigest()[:16]


def workload_key(config, batch, frames, precision):
    """What a PMC pass was measured on / what this run is running: bytes per launch of one workload say nothing about another
    (a batch-8 rehearsal was scored against batch-32 bytes and read 1.004 of the HBM peak)."""
    return f"{config}/batch{batch}/frames{frames}/{precision}"


RUNNING_WORKLOAD = None        # set by main() before any traffic figure is looked up


def pmc_table():
    """(kernel -> entry table of the committed PMC pass that matches the running build AND workload, source) or (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    src = os.path.relpath(files[-1], ROOT)
    with open(files[-1]) as f:
        t = json.load(f)
    if t.get("_build") != kernel_build_id():
        return None, src + " (stale: measured on other kernel sources)"
    if t.get("_workload") == RUNNING_WORKLOAD:
        return {k: v for k, v in t.items() if not k.startswith("_")}, src
    other = (t.get("_passes") or {}).get(RUNNING_WORKLOAD)
    if other:
        return other, src + f" (pass {RUNNING_WORKLOAD})"
    return None, src + f" (no pass for workload {RUNNING_WORKLOAD}; it holds {[t.get('_workload')] + sorted(t.get('_passes') or {})})"


def pmc_traffic(kernel_tag):
    """HBM-side bytes per launch of `kernel_tag` from the last committed PMC pass (tools/pmc_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate rocprofv3 --pmc runs of this same bench, FETCH_SIZE x2 as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the process, so the figure is taken
    from profiles/; None when no pass covers this kernel or when the pass was taken on other kernel sources."""
    t, src = pmc_table()
    if t is None:
        return None, src
    e = t.get(kernel_tag)
    if not e:
        return None, None
    return e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"], src


HBM_FAMILIES = {"winograd_transforms": ("winon_", "wino1d_", "winow_"),
                "pointwise_pool_upcat": ("pointwise_", "avgpool_", "upcat_"),
                "fsq": ("fsq_",)}


def pmc_family_bytes():
    """Per step and per family of HBM-bound kernels (HBM_FAMILIES: kernel-name prefixes), the HBM-side bytes of the last committed
    PMC pass (same file, same source-hash and same-workload rule as pmc_traffic): sum over the family's kernels of bytes per launch x launches per
    step.  The pass's step count comes from its own adam_kernel launches (two per step).  Also the bytes of the two-dimensional
    Winograd layers as a whole - their transforms plus the dense 1x1 instantiations of the GEMM kernels, which only those layers
    launch with more than a few MB - for the wasted-traffic ratio.  (None, source) when there is no valid pass."""
    t, src = pmc_table()
    if t is None:
        return None, src
    adam = t.get("adam_kernel")
    if not adam or not adam.get("launches"):
        return None, src + " (no adam_kernel entry: step count unknown)"
    steps = adam["launches"] / 2.0
    per_step = lambda e: (e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]) * e["launches"] / steps
    out = {fam: 0.0 for fam in HBM_FAMILIES}
    wino2d = 0.0
    for name, e in t.items():
        if name.startswith("_"):
            continue
        for fam, prefixes in HBM_FAMILIES.items():
            if name.startswith(prefixes):
                out[fam] += per_step(e)
        if name.startswith(("winon_", "conv_nt_dense256_kernel")) or (name.startswith(("conv_nt_kernel<", "conv_wgrad_kernel<")) and name.rstrip(">").endswith(", 4")):
            wino2d += per_step(e)
    out["_wino2d_layers"] = wino2d
    return out, src


def hbm_bound_of(tr, batches, lens, steps=2):
    """The HBM-bound half of the step ('HBM roof for FSQ streaming and the fused passes'), measured AFTER the timed
    region on `steps` further optimisation steps: HIP events on the launch stream around every launch of the Winograd transforms
    (through mq_debug_set_pass_hook), the pointwise / pool / upsample-concat passes and the FSQ kernels in situ (ops.pass_timing) give
    ms per step per family; GB per step per family come from the committed PMC pass (null when it was taken on other kernel
    sources, as `traffic`); TB/s and the fraction of the 8 TB/s peak follow.  wasted_traffic_ratio: bytes the two-dimensional
    Winograd layers (refiner 3x3, patch discriminator 5x5 / stride 2) moved - transforms and Winograd-domain GEMMs, PMC - over
    the bytes a fused layer would have to move (every activation tensor a pass reads or writes, once; reported by the library)."""
    from mqgan_amd import ops
    with ops.pass_timing() as t:
        for i in range(steps):
            tr.train_step(batches[i % len(batches)], lens)
        torch.cuda.synchronize()
    summ = t.summary()
    fam_bytes, src = pmc_family_bytes()
    out = {"bound": "hbm", "peak": PEAK_HBM_GBS / 1e3, "unit": "TB/s", "traffic_source": src, "steps_measured": steps, "families": {}}
    for fam in HBM_FAMILIES:
        v = summ.get(fam)
        if not v:
            continue
        ms = v["ms"] / steps
        gb = None if fam_bytes is None else fam_bytes[fam] / 1e9
        out["families"][fam] = {"ms_per_step": ms, "launches_per_step": v["launches"] / steps, "GB_per_step": gb,
                                "achieved": None if gb is None else gb / ms, "frac": None if gb is None else gb / ms / (PEAK_HBM_GBS / 1e3)}
    alg = sum(b for f, b in t.layer_bytes.items() if f in ("winograd2d", "polyphase-winograd2d")) / steps
    out["winograd2d_layers"] = {"algorithmic_GB_per_step": alg / 1e9,
                                "moved_GB_per_step": None if fam_bytes is None else fam_bytes["_wino2d_layers"] / 1e9,
                                "wasted_traffic_ratio": None if (fam_bytes is None or alg <= 0) else fam_bytes["_wino2d_layers"] / alg,
                                "note": "moved: PMC bytes of winon_* and of the dense 1x1 GEMM instantiations; algorithmic: x + y (forward), "
                                        "dy + dx + x (backward) of every such layer, once each"}
    out["all_reduced_form_layers_algorithmic_GB_per_step"] = sum(t.layer_bytes.values()) / steps / 1e9
    return out


def time_events(fn, iters, warm_s=0.25):
    """Mean milliseconds of `iters` back-to-back calls (HIP events) in the STEADY state: the chip's clock ramps over tens of
    milliseconds (tools/microbench/f32_gemm_lab.hip, profiles/r6_f32_gemm_lab.txt: a 0.5 ms kernel measured after one warm-up
    call reads 1.5 GHz, the steady state 2.3-2.4), and these kernels live inside a 190 ms step, so the calls are repeated for
    `warm_s` seconds first."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
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
    # 16384 rows = one config-2 batch (32 x 512 frames): 768 blocks, a single round on 256 CUs - ramp and tail are a third of
    # these 0.1-1.3 ms launches; the last entry shows the same kernel on four batches' worth of rows
    for rows, k, d in ((16384, 1024, 256), (16384, 2048, 512), (16384, 8192, 512), (65536, 8192, 512)):
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
        tag = f"{k}x{d}" + ("" if rows == 16384 else f"_rows{rows}")
        out["l2_argmin_" + tag] = {"n": rows, "ms": ms, "achieved_TFLOPs": flops / (ms / 1e3) / 1e12,
                                   "peak_TFLOPs": PEAK_F32_MFMA_TFLOPS, "frac": flops / (ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                   "bound": "mfma", "algorithmic_GBps": byts / (ms / 1e3) / 1e9,
                                   "kernel": "l2_argmin_kernel: every distance on the f32 MFMA (exact products, k-ordered)"}
        # the same search - same indices and distances, checked here - by bf16 screening + exact f32 re-rank (vq_screen.hip):
        # the 2 N K D distance FLOPs are the ALGORITHMIC work; they run on the bf16 pipe, so the f32-roof fraction may exceed 1
        prep = ops.l2_screen_prepare(cb)
        ws2 = ops._ws_bytes(ops.L().mq_l2_screen_workspace(rows, k, d), x)
        i_ref, d_ref = ops.l2_argmin(x, cb, cn, return_dist=True, workspace=ws)
        i_scr, d_scr = ops.l2_argmin_screened(x, cb, prep, return_dist=True, workspace=ws2)
        same = bool(torch.equal(i_ref, i_scr) and torch.equal(d_ref, d_scr))
        emitted, cap, over = ops.l2_screen_stats(ws2, rows, k, d)
        ms2 = time_events(lambda: ops.l2_argmin_screened(x, cb, prep, workspace=ws2), 10)
        # roofline of the screened search = the pipe it runs on: the 2 N K D distance flops are formed ONCE on the bf16 MFMA (r5;
        # r3-r4 formed them twice), so `frac` is against the bf16 peak; the ratio to the f32 kernel is a speed-up, not a
        # roofline fraction
        out["l2_argmin_screened_" + tag] = {"n": rows, "ms": ms2, "algorithmic_TFLOPs": flops / (ms2 / 1e3) / 1e12,
                                            "peak_TFLOPs": PEAK_BF16_MFMA_TFLOPS,
                                            "frac": flops / (ms2 / 1e3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                            "executed_bf16_TFLOPs": flops / (ms2 / 1e3) / 1e12, "launches": 2 if d <= 256 else 3,
                                            "speedup_vs_f32_kernel": ms / ms2, "identical_to_f32_kernel": same,
                                            "list_entries_per_row": emitted / rows, "list_capacity_per_row": cap / rows,
                                            "rows_searched_exhaustively": over,
                                            "algorithmic_GBps": byts / (ms2 / 1e3) / 1e9, "bound": "mfma (bf16 pipe)",
                                            "kernel": ("l2_search_kernel (d <= 256: the block's rows rounded to bf16 into an LDS image and bounded from their "
                                                       "measured rounding residuals in the kernel, code tiles streamed through a four-stage LDS-DMA ring, "
                                                       "scores formed once on the bf16 MFMA, candidate windows in registers) + l2_rerank2_kernel (windows "
                                                       "against the final maximum, exact k-ordered f32 chain out of LDS-DMA staged rows, idx / distance "
                                                       "written)" if d <= 256 else
                                                       "l2_screen_prep_kernel (x -> bf16, per-row error bound from the measured rounding residuals) + "
                                                       "l2_screen_kernel (bf16 MFMA scores formed once, code norm as the accumulator seed, candidates within "
                                                       "the bound of the running row maximum into private lists) + l2_rerank_kernel (lists filtered against "
                                                       "the final maximum, exact k-ordered f32 chain out of LDS-DMA staged rows, idx / distance written)")
                                                      + "; codebook prepared once, x prepared inside the timed call"}
    return out


def _oracle_trainer(cfg):
    from oracle import train_step as TS
    Pg = TS.make_leaves(TS.init_generator_params(cfg))
    pc, mc = cfg["model"]["discriminator_patch"], cfg["model"]["discriminator_multibin"]
    Pp = TS.make_leaves(TS.init_patch_params(80, pc["hidden_channels"], pc["kernel_sizes"]))
    Pm = TS.make_leaves(TS.init_multibin_params(80, mc["n_bins"], mc["hidden_channels"], mc["kernel_sizes"]))
    return TS.OracleTrainer(Pg, Pp, Pm, cfg, gen_training=True)


def cpu_baseline(seconds_budget=30.0):
    """The oracle's train step on the host cores: hifispeech architecture, mel 80, batch 2 x 256 frames (BASELINE
    config 1), discriminators active, generator in train mode.  The thread count is picked by a short sweep on a
    small probe step (more threads than the cgroup's share of cores only adds contention), then one untimed warm-up
    step of the SAME shape and at least three timed steps."""
    from mqgan_amd.train import synthetic_mel
    cfg = load_config("hifispeech")
    B, T = 2, 256

    def batch(n0, b, t):
        return (torch.from_numpy(np.stack([synthetic_mel(t, 80, n0 + i) for i in range(b)])), torch.tensor([t] * b, dtype=torch.int32))

    ncpu = os.cpu_count() or 8
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    cands = sorted({c for c in (8, 16, 32, 64, ncpu // 2) if 1 <= c <= ncpu} | {min(8, ncpu)})
    tr = _oracle_trainer(cfg)
    probe = {}
    xs, ls = batch(900, 1, 64)
    torch.set_num_threads(cands[0])
    tr.step(xs, ls)                                   # page everything in
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.time()
        tr.step(xs, ls)
        probe[c] = time.time() - t0
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    tr = _oracle_trainer(cfg)                         # fresh optimiser state for the measured run
    x, l = batch(0, B, T)
    t0 = time.time()
    tr.step(x, l)                                     # warm-up: same shape as the timed steps
    warm = time.time() - t0
    n, t0, per_step = 0, time.time(), []
    while n < 9:                                      # 10 steps, the first discarded
        x, l = batch((n + 1) * B, B, T)
        t1 = time.time()
        tr.step(x, l)
        per_step.append(time.time() - t1)
        n += 1
        el = time.time() - t0
        if n >= 3 and el + el / n > seconds_budget:  # ... bounded by the bench contract's 10-30 s of CPU work
            break
    med = sorted(per_step)[len(per_step) // 2]
    return {"value": B * T / med, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"median of {n} timed steps (BASELINE.md section 3 asks for 9 after one discarded step; the bench contract bounds "
                      f"the CPU leg to about {seconds_budget:.0f} s, which stops it at {n}; per-step seconds "
                      + "/".join(f"{v:.2f}" for v in per_step) + f"; mean rate {n * B * T / el:.1f} frames/s) after a same-shape warm-up "
                      f"step of {warm:.1f} s, batch {B} x {T} frames, hifispeech@80, D active, fp32, oracle/train_step.py, {el:.1f} s on "
                      f"{threads} threads; thread sweep on a 1 x 64 probe step: "
                      + ", ".join(f"{c}: {v:.2f} s" for c, v in sorted(probe.items()))
                      + f" ({os.cpu_count()} logical CPUs visible); reference's own figure on 8 threads: 38 frames/s (BASELINE.md)"}


def quick_engine(cfg, precision, device, batches, lens, B, T, steps=4, warmup=2):
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
                          "bf16": "bf16 MFMA operands, f32 accumulate; refiner / discriminator activations stored as bf16 tensors "
                                  "(reference autocast mode), elsewhere f32 tensors rounded in-kernel",
                          "fp32": "exact f32 products on the f32 MFMA, every conv on the direct implicit-GEMM kernels "
                                  "(Winograd switched off for this run)"}[precision]}
    del tr
    torch.cuda.empty_cache()
    return out


def peak_of(tag):
    """Dense MFMA peak of the engine a conv kernel instantiation runs on (MI355X_MICROARCH.md)."""
    if "_x3_" in tag:
        return PEAK_BF16_MFMA_TFLOPS / 6      # the split engine spends 6 bf16 MFMAs per f32 product block
    if "bf16" in tag or "_h2" in tag or "_h_" in tag:          # conv_h2_*, conv_h2s_*, conv_h2s96*_kernel, conv_wgrad_h2*
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_F32_MFMA_TFLOPS


def roofline_of(timer, steps, elapsed, step_peak):
    """Dominant-kernel roofline (live HIP-event timing of every conv launch of the timed region) plus the whole-step
    figure: executed conv FLOP / wall time against the engine's peak."""
    summ = timer.summary()
    total_conv_ms = sum(v["ms"] for v in summ.values())
    total_flop = sum(v["flops"] for v in summ.values())
    dom_tag, dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
    ach = dom["flops"] / (dom["ms"] / 1e3) / 1e12
    traffic, traffic_src = pmc_traffic(dom_tag)
    peak = peak_of(dom_tag)
    step_tf = total_flop / elapsed / 1e12
    return {"bound": "mfma", "kernel": dom_tag, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_flop_per_launch": dom["flops"] / dom["launches"], "launches_per_step": dom["launches"] / steps,
            "avg_launch_ms": dom["ms"] / dom["launches"], "share_of_step_time": dom["ms"] / 1e3 / elapsed,
            "step": {"flop_executed_per_step": total_flop / steps, "TFLOP/s": step_tf, "peak": step_peak, "frac": step_tf / step_peak,
                     "note": "conv / linear FLOP actually executed (all timed launches) over wall time of the timed steps"},
            "all_conv_kernels": {k: {"ms_per_step": v["ms"] / steps, "TFLOPs": v["flops"] / (v["ms"] / 1e3) / 1e12,
                                     "launches_per_step": v["launches"] / steps} for k, v in sorted(summ.items())},
            "conv_share_of_step_time": total_conv_ms / 1e3 / elapsed}


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL's version banner and gloo's connection notes are printed by native code
    straight to file descriptor 1, so from here on descriptor 1 is stderr for everybody (Python included) and the JSON line goes
    to a private duplicate of the original stdout (emit)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def _trace(rank, msg):
    """MQGAN_BENCH_TRACE=1: progress lines on stderr (where does a multi-rank run stop?)."""
    if os.environ.get("MQGAN_BENCH_TRACE") == "1":
        print(f"[bench rank {rank}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)


def host_ms_per_step(tr, batches, lens, samples=3):
    """CPU time the host needs to ISSUE one optimisation step (Python + ctypes launches + hooks + collectives), measured from
    an idle stream to the return of train_step with the stream left unsynchronised: as long as this stays below the GPU's
    ms_per_step the stream never runs dry.  Median of `samples`; the steps run after the timed region."""
    out = []
    for i in range(samples):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_step(batches[i % len(batches)], lens)
        out.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    return sorted(out)[len(out) // 2]


def collectives_of(tr, world, forced):
    """What the step exchanged (rank 0's view)."""
    if not (world > 1 or forced):
        return None
    out = dict(tr.bench_comm or {})          # exposed communication and per-rank step times of the timed region (timed_run)
    abi = bool(getattr(tr.comm, "buckets_async", False))
    out.update({"backend": "rccl behind the C ABI (mq_allreduce_bucket)" if abi else dist.get_backend(), "ranks_counted_by_all_reduce": world, "forced_on_one_rank": bool(forced and world == 1),
            "gradient_buckets_per_step": len(tr.reducer_g.buckets) + len(tr.reducer_d.buckets),
            "bucket_MiB": [round((e - s) * 4 / 2**20, 1) for red in (tr.reducer_d, tr.reducer_g) for (s, e, _) in red.buckets],
            # the bucket whose gradients are ready LAST (the first layers: backward has ended, nothing is left to hide its
            # all-reduce behind); buckets taper towards it (mqgan_amd/ddp.py)
            "last_bucket_MiB": {"discriminators": round(tr.reducer_d.last_bucket_bytes / 2**20, 2),
                                "generator": round(tr.reducer_g.last_bucket_bytes / 2**20, 2)},
            "asynchronous": not tr.reducer_g._blocking, "host_drains_stream_before_gloo_device_collective": tr.reducer_g._drain_first,
            "loss_sum_all_reduces_total": getattr(tr.comm, "calls", None),
            "exposed_comm_note": "HIP events on the compute stream around every wait for a collective (the blocking loss-sum "
                                 "all-reduces; the waits on the asynchronous gradient buckets after backward): time the step "
                                 "could not hide behind compute; max over ranks"})
    return out


def timed_run(cfg, precision, B, T, steps, warmup, device, world, rank, force=False, abi=False):
    """W untimed + K timed optimisation steps of one configuration; returns the trainer, inputs, kernel timer, wall
    seconds of the timed region (max over ranks) and the last step's losses."""
    import copy
    from mqgan_amd import ops
    from mqgan_amd.train import Trainer
    cfg = copy.deepcopy(cfg)
    cfg["training"]["precision"] = precision
    mel = cfg["model"]["mel_channels"]
    _trace(rank, "building the trainer")
    comm = None
    if abi:                                  # device collectives through include/mqgan_hip.h's own RCCL entry points
        from mqgan_amd.comm import RcclAbiComm
        comm = RcclAbiComm(force=force)
    tr = Trainer(cfg, device=device, build_data=False, comm=comm, force_collectives=force)
    tr.generator.train()
    tr.patch_discriminator.train()
    tr.multibin_discriminator.train()
    tr.epoch = 1
    batches = synthetic_batches(B, T, mel, 4, device, rank)
    lens = torch.full((B,), T, dtype=torch.int32, device=device)
    for i in range(warmup):
        _trace(rank, f"warm-up step {i}")
        tr.train_step(batches[i % len(batches)], lens)
        if os.environ.get("MQGAN_BENCH_TRACE") == "1":
            torch.cuda.synchronize()
    _trace(rank, "warm-up done")
    timer = ops.KernelTimer()
    ops.TIMER = timer
    clock = getattr(tr.comm, "clock", None)          # HIP events around every point where the compute stream waits for a collective
    if clock is not None:
        clock.read_ms()
        clock.enabled = True
    calls0 = getattr(tr.comm, "calls", 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = tr.train_step(batches[i % len(batches)], lens)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                   # this rank's own K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    tr.bench_comm = None
    if clock is not None:
        clock.enabled = False
        exposed = clock.read_ms() / steps
        per_rank = torch.tensor([own / steps * 1e3, exposed], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(per_rank) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, per_rank)
        else:
            allr = [per_rank]
        ms = [float(t[0]) for t in allr]
        ex = [float(t[1]) for t in allr]
        tail = torch.tensor([tr.reducer_d.tail_ms(), tr.reducer_g.tail_ms()], device=device, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tail, op=dist.ReduceOp.MAX)
        tr.bench_comm = {"loss_sum_all_reduces_per_step": (getattr(tr.comm, "calls", 0) - calls0) / steps,
                         # compute-stream time from the launch of a step's last gradient bucket to the end of the waits in
                         # front of clip / Adam (mean over the timed steps, max over ranks): the tail no backward can hide
                         "ms_between_last_bucket_ready_and_clip": {"discriminators": float(tail[0]), "generator": float(tail[1])},
                         "exposed_comm_ms_per_step": max(ex), "exposed_comm_ms_per_step_by_rank": [round(v, 3) for v in ex],
                         "rank_ms_per_step_min": min(ms), "rank_ms_per_step_max": max(ms),
                         "rank_ms_per_step": [round(v, 3) for v in ms]}
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return tr, batches, lens, timer, elapsed, last


def config3_line(device):
    """BASELINE config 3 (hifimusic, 16 x 128 mels x 1024 frames, bf16 convs / fp32 VQ) on this GPU: 3 timed steps after
    two warm-up steps, with its own dominant-kernel and whole-step roofline.  Reported under `configs`, never as `value`."""
    cfg = load_config("hifimusic")
    B, T = cfg["data"]["batch_size"], cfg["data"]["crop_len"]
    steps, warmup = 3, 2          # two warm-up steps: the caching allocator has to grow to this config's footprint first
    tr, batches, lens, timer, elapsed, last = timed_run(cfg, "bf16", B, T, steps, warmup, device, 1, 0)
    global RUNNING_WORKLOAD
    keep, RUNNING_WORKLOAD = RUNNING_WORKLOAD, workload_key("hifimusic", B, T, "bf16")
    try:
        roof = roofline_of(timer, steps, elapsed, PEAK_BF16_MFMA_TFLOPS)
    finally:
        RUNNING_WORKLOAD = keep
    out = {"workload": f"hifimusic PreEncoder VQ-GAN train step, {cfg['model']['mel_channels']} mels x {T} frames, batch {B}, "
                       "bf16 convolutions with bf16 activation storage (refiner, discriminators), f32 accumulation, f32 VQ / losses / "
                       "optimiser", "dtype": "bf16 convs / f32 VQ", "value": B * T * steps / elapsed, "unit": "mel-frames/s",
           "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
           "losses_finite": all(np.isfinite(float(v)) for v in last.values()),
           "roofline": roof}
    del tr, batches
    torch.cuda.empty_cache()
    return out


def forced_collectives_line(cfg, device, B, T, ms_plain):
    """The cost of the N > 1 code path in the driver's N = 1 record: one forced RCCL rank - process group
    of one, every gradient bucket all-reduced asynchronously on RCCL's own stream from the post-accumulate hooks, the three
    loss-sum all-reduces per step - for 3 timed steps of the SAME workload after the main timed region.  Never `value`."""
    steps, warmup = 3, 1
    if not dist.is_initialized():
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        dist.init_process_group("nccl", device_id=device)
    try:
        tr, batches, lens, timer, elapsed, last = timed_run(cfg, "fp32", B, T, steps, warmup, device, 1, 0, force=True)
        out = collectives_of(tr, 1, True) or {}
        out.update({"ms_per_step": elapsed / steps * 1e3, "ms_per_step_without_collectives": ms_plain, "steps": steps, "warmup": warmup,
                    "losses_finite": all(np.isfinite(float(v)) for v in last.values())})
        del tr, batches
    finally:
        dist.destroy_process_group()
    torch.cuda.empty_cache()
    # the same three steps with the collectives behind the C ABI (mq_comm_init / mq_allreduce_bucket, no torch.distributed)
    try:
        # two warm-up steps: the allocator regrows after empty_cache() and the first step of an epoch is the long one (train-mode
        # discriminators); with one, this block read 262 ms where the same trainer then runs at the plain step's rate
        tr, batches, lens, timer, elapsed, last = timed_run(cfg, "fp32", B, T, steps, 2, device, 1, 0, force=True, abi=True)
        out["through_the_c_abi"] = {"ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": 2, "exposed_comm_ms_per_step": (tr.bench_comm or {}).get("exposed_comm_ms_per_step"),
                                    "ms_between_last_bucket_ready_and_clip": (tr.bench_comm or {}).get("ms_between_last_bucket_ready_and_clip"),
                                    "collective_calls": tr.comm.calls, "losses_finite": all(np.isfinite(float(v)) for v in last.values()),
                                    "entry_points": "mq_comm_unique_id / mq_comm_init / mq_allreduce_bucket / mq_comm_destroy"}
        tr.comm.close()
        del tr, batches
    except Exception as e:                   # noqa: BLE001 - a secondary block must not take the line down
        out["through_the_c_abi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising HIP (the launcher parent must stay free of GPU state: on this ROCm
    build torch.cuda.device_count() falls back to hipGetDeviceCount, which opens KFD): the KFD topology lists one node per
    agent, GPUs are the ones with SIMDs; HIP_/ROCR_/CUDA_VISIBLE_DEVICES narrow it.  No KFD at all = no GPU; None when the
    topology exists but cannot be read (the ranks then find out themselves: set_device fails loudly)."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return 0
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher in the environment: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on 127.0.0.1) as a CHILD of this process - which makes no HIP call, not even a device count (see
    visible_gpu_count) - pass rank 0's JSON line through on stdout and exit with the child's code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MQGAN_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    import __graft_entry__ as ge
    ge.build_native()        # compile BEFORE the ranks exist (hipcc only: this parent still makes no HIP call)
    return subprocess.call(cmd, env=env)


def reencode_line(n_files=128):
    """SURVEY 8f N2 on this GPU: the bulk re-encoding job (ScriptedPreEncoder encode -> decode of a .npy tree, hifispeech
    architecture, random weights, fp32, lengths U(200, 1200), batch 32) under the reference's batching ('order', the default)
    and the length-sorted policy.  File I/O included (that is the job); never `value`."""
    import contextlib
    import io
    import tempfile
    from mqgan_amd.export import export
    from mqgan_amd.preencoder import PreEncoder
    from mqgan_amd.reencode_spectrograms import reencode_spectrograms
    from mqgan_amd.train import synthetic_mel
    cfg = load_config("hifispeech")
    m, g = cfg["model"], cfg["model"]["generator"]
    out = {"workload": f"re-encode {n_files} synthetic .npy files of 200-1200 frames x {m['mel_channels']} mels, hifispeech PreEncoder "
                       "(export -> ScriptedPreEncoder.encode -> .decode), fp32, batch 32, file I/O included", "unit": "mel-frames/s"}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(0)
        model = PreEncoder(m["mel_channels"], g["channels"], g["kernel_sizes"], fsq_levels=g["fsq_levels"], dropout=0.0,
                           refiner_base_channels=g["refiner_base_channels"], refiner_depth=g["refiner_depth"])
        ck, cf = os.path.join(tmp, "ck.pth"), os.path.join(tmp, "cfg.yaml")
        torch.save({"model_state_dict": model.state_dict()}, ck)
        with open(cf, "w") as f:
            yaml.safe_dump(cfg, f)
        mdir = export(ck, cf, os.path.join(tmp, "model"))
        src = os.path.join(tmp, "in")
        os.makedirs(src)
        rng = np.random.default_rng(0)
        for i in range(n_files):
            np.save(os.path.join(src, f"{i:05d}.npy"), synthetic_mel(int(rng.integers(200, 1201)), m["mel_channels"], i))
        for tag, bucket in (("warm", "sorted"), ("order", "order"), ("sorted", "sorted")):      # first pass: allocator / kernels warm
            st = {}
            reencode_spectrograms(mdir, src, os.path.join(tmp, "out_" + tag), "cuda", 32, bucket=bucket, stats=st)
            if tag != "warm":
                out[tag] = {"value": st["frames_per_s"], "files": st["files"], "frames": st["frames"],
                            "padding_fraction": st["pad_fraction"], "seconds": st["seconds"]}
    return out


def loader_line(device, n_files=576, file_frames=640):
    """The same config-2 step fed by the real input pipeline instead of resident batches (the reference's loop
    train.py:511-537): synthetic .npy files on disk -> DataLoader (num_workers 0, random 512-frame crops, collate) -> pinned
    staging -> H2D on a side stream (PrefetchLoader) -> train_step.  Trainer._train_epoch itself is timed: one warm-up epoch,
    one timed epoch of 18 steps in the steady state of a run (r6: the loader lives across epochs, the first batches of an epoch are
    staged during the last steps of the one before; the discriminators' first step of every epoch runs in train mode - one power
    iteration, real and fake as separate batches - as in the reference, train.py:505-506).  The PCIe-inclusive rate - reported
    beside `value`, never as `value`."""
    import copy
    import tempfile
    from mqgan_amd.train import Trainer, synthetic_mel
    cfg = copy.deepcopy(load_config("hifispeech"))
    B, T, mel = cfg["data"]["batch_size"], cfg["data"]["crop_len"], cfg["model"]["mel_channels"]
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n_files):
            np.save(os.path.join(tmp, f"{i:05d}.npy"), synthetic_mel(file_frames, mel, 50000 + i))
        cfg["data"]["data_dir"] = tmp
        cfg["data"]["output_dir"] = None
        cfg["data"]["validation_split"] = 0.0
        tr = Trainer(cfg, device=device, build_data=True)
        steps = len(tr.train_loader)
        tr._train_epoch(1, next_epoch=2)             # warm-up: allocator, kernels, page cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr._train_epoch(2, next_epoch=3)             # a steady-state epoch: its first batches were staged during epoch 1's last
        torch.cuda.synchronize()                     # steps, and it starts epoch 3's producer during its own
        dt = time.perf_counter() - t0
        tr._prefetch.close()
        del tr
    torch.cuda.empty_cache()
    return {"workload": f"hifispeech train step fed by DataLoader + PrefetchLoader: {n_files} .npy files of {file_frames} x {mel}, "
                        f"random crops of {T}, batch {B}, num_workers 0 (as the reference), pinned staging, H2D on a side stream",
            "value": steps * B * T / dt, "unit": "mel-frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps}


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
    ap.add_argument("--no-configs", action="store_true", help="skip the 3-step BASELINE config-3 (hifimusic, bf16) line")
    ap.add_argument("--cpu-seconds", type=float, default=70.0,
                    help="bound of the CPU baseline's timed steps: 70 s lets a 16-thread host take the 9 steps BASELINE.md section 3 "
                         "asks for (about 6.5 s each); it stops earlier, at >= 3 steps, when they would not fit")
    ap.add_argument("--force-collectives", action="store_true",
                    help="issue every data-parallel collective (gradient buckets on RCCL's side stream, loss sums) even with one "
                         "rank: the N = 1 cost of the N > 1 code path")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous rehearsal: start the ranks, count them with an all-reduce, print the line with "
                         "value null - no kernels (runs on a box without GPUs when MQGAN_BENCH_BACKEND=gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE"))
    if not launched and args.gpus > 1:
        # no launcher: this process becomes one (before anything here touches a GPU) and relays the ranks' output
        backend = os.environ.get("MQGAN_BENCH_BACKEND", "nccl")
        have = visible_gpu_count()
        if backend == "nccl" and os.environ.get("MQGAN_BENCH_SHARE_GPU") != "1" and have is not None and have < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible")
        raise SystemExit(self_launch(args.gpus))

    claim_stdout()          # this process is a rank: native libraries must not get a word in on the JSON line's channel
    import __graft_entry__ as ge
    if os.environ.get("MQGAN_BENCH_TRACE") == "1":
        # a stuck rank leaves the Python stacks of all its threads and the state / kernel wait channel of every OS thread
        # (the collective library's workers have no Python frames) in MQGAN_BENCH_TRACE_DIR
        import faulthandler
        import threading
        after = int(os.environ.get("MQGAN_BENCH_TRACE_AFTER", "150"))
        faulthandler.dump_traceback_later(after, repeat=False, file=sys.stderr)

        def _dump():
            time.sleep(after)
            d = os.environ.get("MQGAN_BENCH_TRACE_DIR", os.path.join(ROOT, "gpurun_out"))
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, f"stuck_rank{os.environ.get('RANK', '0')}.txt"), "w") as f:
                f.write(f"bench.py still running after {after} s\n== OS threads ==\n")
                for tid in sorted(os.listdir("/proc/self/task"), key=int):
                    def rd(n):
                        try:
                            with open(f"/proc/self/task/{tid}/{n}") as fh:
                                return fh.read().strip()
                        except OSError:
                            return "?"
                    st = rd("stat")
                    f.write(f"tid {tid} {rd('comm'):<18} state {st.rsplit(')', 1)[-1].split()[0] if ')' in st else '?'} wchan {rd('wchan')}\n")
                f.write("== Python stacks ==\n")
                f.flush()
                faulthandler.dump_traceback(file=f, all_threads=True)
        threading.Thread(target=_dump, daemon=True, name="bench-trace").start()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world_env != args.gpus:
        # never a silent run on another number of GPUs than the one asked for
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world_env}")
    # MQGAN_BENCH_SHARE_GPU=1 (rehearsal only): all ranks on device 0 with the gloo backend, to exercise the N > 1
    # code path on a one-GPU box; RCCL needs one device per rank.
    share = os.environ.get("MQGAN_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("MQGAN_BENCH_BACKEND", "gloo" if share else "nccl")
    if share:
        local = 0
    use_pg = world_env > 1 or args.force_collectives
    if use_pg and not launched:                      # one forced rank without a launcher: rendezvous with ourselves
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    if args.dry_run:
        # rendezvous + rank count only: what the driver's `--gpus N` command has to get right before any kernel runs
        world = 1
        if use_pg:
            if backend == "nccl":
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
                ones = torch.ones(1, device=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
                ones = torch.ones(1)
            dist.all_reduce(ones)
            world = int(ones.item())
            if world != dist.get_world_size():
                raise SystemExit(f"all-reduce of ones counted {world} ranks, the process group says {dist.get_world_size()}")
        if rank == 0:
            emit({"metric": METRIC, "value": None, "unit": "mel-frames/s", "n_gpus": world, "steps": 0, "warmup": 0,
                  "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dry_run": True, "backend": backend if use_pg else None,
                  "self_launched": os.environ.get("MQGAN_BENCH_SELF_LAUNCHED") == "1"})
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        return
    # every rank builds (a no-op `make` behind a file lock once the first one is through): nobody waits in the rendezvous below
    # while rank 0 alone compiles - r4 relied on init_process_group's 10-minute default timeout for a cold 2-minute build
    ge.build()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    world = 1
    if use_pg:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # the rank count the line reports is the one the collective library saw, not the flag
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        world = int(ones.item())
        if world != dist.get_world_size() or world != args.gpus:
            raise SystemExit(f"all-reduce of ones counted {world} ranks; process group {dist.get_world_size()}, --gpus {args.gpus}")
        dist.barrier()
    from mqgan_amd import _lib, ops
    from mqgan_amd.train import Trainer
    _lib.load()                                   # fails loudly if the HIP library is missing

    cfg = load_config(args.config)
    B = args.batch or cfg["data"]["batch_size"]
    T = args.frames or cfg["data"]["crop_len"]
    mel = cfg["model"]["mel_channels"]
    global RUNNING_WORKLOAD
    RUNNING_WORKLOAD = workload_key(args.config, B, T, args.precision)
    tr, batches, lens, timer, elapsed, last = timed_run(cfg, args.precision, B, T, args.steps, args.warmup, device, world, rank,
                                                        force=args.force_collectives)
    host_ms = host_ms_per_step(tr, batches, lens)
    finite = all(np.isfinite(float(v)) for v in last.values())
    hbm = hbm_bound_of(tr, batches, lens)          # every rank steps (the step holds collectives); rank 0 reports

    if rank == 0:
        frames = world * B * T * args.steps
        step_peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "fp32x3": PEAK_BF16_MFMA_TFLOPS / 6}[args.precision]
        roof = roofline_of(timer, args.steps, elapsed, step_peak)
        roof["hbm_bound"] = hbm
        roof["wasted_traffic_ratio"] = (hbm.get("winograd2d_layers") or {}).get("wasted_traffic_ratio")
        line = {"metric": METRIC, "value": frames / elapsed, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 convs (f32 accumulate, bf16 activation storage) / f32 VQ",
                                              "fp32x3": "f32 via 3-term bf16 split (6 products, f32 accumulate) / f32 VQ"}[args.precision],
                "data": "synthetic",
                "config": {"workload": f"{args.config} PreEncoder VQ-GAN train step (G fwd + D step + G step + clip/Adam), "
                                       f"{mel} mels x {T} frames, batch {B}/GPU, FSQ {cfg['model']['generator']['fsq_levels']}, "
                                       "D active, FM off, train-mode dropout", "workload_key": RUNNING_WORKLOAD, "global_batch": world * B, "frames": T,
                           "parallelism": f"dp{world}",
                           "conv_algorithm": ("implicit GEMM; where a cheaper exact-f32 form exists, Winograd with f32 transforms, products on the same "
                                              "MFMA engine and f32 accumulation: 3x3 / stride-1 layers as F(4x5,3x3) / F(4x4) / F(3x3) (tile per "
                                              "image), the patch discriminator's 5x5 / stride-2 layers as the same over the polyphase components "
                                              "of their input and its 3x3 / stride-(2,1) layers as F(4,3) along time, the multi-bin "
                                              "discriminator's (3 x k) layers as F(5,3) along the bins (its (3 x 5) layer nested with F(4,5) along "
                                              "time), the encoder / decoder 1-D convolutions as F(4,3) / F(4,5) / F(2,7) "
                                              "(MQGAN_WINO_MINC=0 / MQGAN_WINO_S2D_MINC=0 / MQGAN_WINO1D_MINC=0 / MQGAN_WINOW_MINC=0 switch them off)") if args.precision != "bf16"
                           else "implicit GEMM on bf16-storage kernels"},
                "losses_finite": bool(finite), "roofline": roof,
                "host_ms_per_step": host_ms, "collectives": collectives_of(tr, world, args.force_collectives)}
        del tr
        torch.cuda.empty_cache()
        if world == 1 and not args.no_alt and args.precision == "fp32":
            # secondary figures, same workload, opt-in engines (never `value`): see DESIGN.md section 3
            line["other_engines"] = {p: quick_engine(cfg, p, device, batches, lens, B, T) for p in ("fp32x3", "bf16")}
            from mqgan_amd import ops as _ops
            keep, _ops.WINO_MIN_CHANNELS = _ops.WINO_MIN_CHANNELS, 0          # every layer on the direct kernels (A/B, same box)
            keep1, _ops.WINO1D_MIN_CHANNELS = _ops.WINO1D_MIN_CHANNELS, 0
            keep2, _ops.WINO_S2D_MIN_CHANNELS = _ops.WINO_S2D_MIN_CHANNELS, 0
            keep3, _ops.WINOW_MIN_CHANNELS = _ops.WINOW_MIN_CHANNELS, 0
            try:
                line["other_engines"]["fp32_direct"] = quick_engine(cfg, "fp32", device, batches, lens, B, T)
            finally:
                _ops.WINO_MIN_CHANNELS, _ops.WINO1D_MIN_CHANNELS, _ops.WINO_S2D_MIN_CHANNELS, _ops.WINOW_MIN_CHANNELS = keep, keep1, keep2, keep3
        if world == 1 and not args.no_configs and args.config == "hifispeech" and args.precision == "fp32":
            del batches
            torch.cuda.empty_cache()
            line["configs"] = {"config3_hifimusic_bf16": config3_line(device)}
            torch.cuda.empty_cache()
            line["configs"]["reencode_n2_hifispeech_fp32"] = reencode_line()
            torch.cuda.empty_cache()
            line["configs"]["config2_through_the_loader"] = loader_line(device)
        if world == 1 and not use_pg and not args.no_configs and args.config == "hifispeech" and args.precision == "fp32":
            try:
                line["collectives_forced_1rank"] = forced_collectives_line(cfg, device, B, T, line["ms_per_step"])
            except Exception as e:          # a box without a usable RCCL must not cost the line its headline
                line["collectives_forced_1rank"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_vq:
            line["vq"] = bench_vq(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        # the figures a reader of a TRUNCATED line needs, right behind the contract's keys: the secondary configurations' step
        # times and the dominant kernel's roofline fraction
        summary = {"roofline_frac": (line.get("roofline") or {}).get("frac"), "step_frac_of_peak": ((line.get("roofline") or {}).get("step") or {}).get("frac")}
        c3 = (line.get("configs") or {}).get("config3_hifimusic_bf16")
        if c3:
            summary["config3_hifimusic_bf16_ms_per_step"] = c3["ms_per_step"]
            summary["config3_roofline_frac"] = (c3.get("roofline") or {}).get("frac")
        for k, v in (line.get("other_engines") or {}).items():
            summary[k + "_ms_per_step"] = v.get("ms_per_step")
        if line.get("cpu_baseline"):
            summary["cpu_baseline_frames_per_s"] = line["cpu_baseline"].get("value")
        if line.get("collectives"):
            summary["exposed_comm_ms_per_step"] = line["collectives"].get("exposed_comm_ms_per_step")
        fc = line.get("collectives_forced_1rank") or {}
        if "ms_per_step" in fc:
            summary["forced_1rank_collectives"] = {k: fc.get(k) for k in ("ms_per_step", "exposed_comm_ms_per_step", "ms_between_last_bucket_ready_and_clip", "bucket_MiB")}
        ld = (line.get("configs") or {}).get("config2_through_the_loader")
        if ld:
            summary["config2_through_the_loader_ms_per_step"] = ld.get("ms_per_step")
        for k, v in (line.get("vq") or {}).items():
            if k.startswith("l2_argmin_screened_") and "rows" not in k:
                summary[k + "_us"] = round(v["ms"] * 1e3, 1)
        front = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
        ordered = {k: line[k] for k in front if k in line}
        ordered["summary"] = summary
        ordered.update({k: v for k, v in line.items() if k not in ordered})
        ordered["summary_tail"] = summary           # the same figures as the LAST key: a reader who only keeps the tail of the line
        emit(ordered)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
