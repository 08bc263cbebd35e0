#!/usr/bin/env python3
"""Twenty optimisation steps of BASELINE config 2 (full size, train-mode dropout, discriminators active) twice from the same
weights, batches and dropout seeds: every convolution on the direct implicit-GEMM kernels, and with the Winograd forms that are
the default - and, as the yardstick, the direct kernels again on the fp32x3 engine (the same algorithm, f32
accuracy, another order of roundings).  Prints the losses step by step and the relative difference of each run to the first: how
far f32 rounding differences of 1e-6 per layer move a GAN training trajectory whose first Adam steps are +-lr per weight."""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mqgan_amd import ops
from mqgan_amd.train import Trainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = bench.load_config("hifispeech")
B, T, mel = cfg["data"]["batch_size"], cfg["data"]["crop_len"], cfg["model"]["mel_channels"]
batches = bench.synthetic_batches(B, T, mel, 4, dev, 0)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
lens[3], lens[17] = 300, 411
runs = {}
import copy
for name, minc, prec in (("direct", 0, "fp32"), ("winograd", 128, "fp32"), ("direct_x3", 0, "fp32x3")):
    ops.WINO_MIN_CHANNELS = ops.WINO1D_MIN_CHANNELS = ops.WINO_S2D_MIN_CHANNELS = ops.WINOW_MIN_CHANNELS = minc
    torch.manual_seed(42)
    ops.manual_seed(42)
    c = copy.deepcopy(cfg)
    c["training"]["precision"] = prec
    tr = Trainer(c, device=dev, build_data=False)
    tr.generator.train(); tr.patch_discriminator.train(); tr.multibin_discriminator.train()
    tr.epoch = 1
    hist = []
    for i in range(STEPS):
        out = tr.train_step(batches[i % len(batches)], lens)
        hist.append({k: float(v) for k, v in out.items() if k.startswith("loss")})
    runs[name] = hist
    del tr
    torch.cuda.empty_cache()
keys = [k for k in runs["direct"][0] if k in ("loss_d", "loss_g_total", "loss_recon_post", "loss_gan")]
print("per loss: direct f32 | winograd f32 (rel. diff) | direct fp32x3 (rel. diff)")
print("step  " + "  ".join(f"{k:>58s}" for k in keys))
worst = {"winograd": 0.0, "direct_x3": 0.0}
for i in range(STEPS):
    cells = []
    for k in keys:
        a = runs["direct"][i][k]
        cell = f"{a:11.6f}"
        for other in ("winograd", "direct_x3"):
            b = runs[other][i][k]
            rel = abs(a - b) / max(abs(a), 1e-12)
            worst[other] = max(worst[other], rel)
            cell += f" | {b:11.6f} ({rel:7.1e})"
        cells.append(cell)
    print(f"{i:4d}  " + "  ".join(cells))
print(f"largest relative difference of a loss to the direct f32 run over {STEPS} steps: winograd {worst['winograd']:.2e}, "
      f"direct kernels on fp32x3 {worst['direct_x3']:.2e}")
