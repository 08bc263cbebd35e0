#!/usr/bin/env python3
"""Soak of the whole training loop with the reference's crop-length LIST (train.py:140-197: a different T per batch, so
the allocator hands out different gradient addresses every step): hifispeech architecture, synthetic .npy tree,
DataLoader + PrefetchLoader, several epochs per precision; prints the loss history (must stay finite and the reconstruction
losses must fall).  usage: python tools/soak_loader.py [fp32] [bf16] [fp32x3]"""
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.train import Trainer, synthetic_mel  # noqa: E402

with open(os.path.join(ROOT, "configs", "model_config_hifispeech.yaml")) as f:
    base = yaml.safe_load(f)
with tempfile.TemporaryDirectory() as tmp:
    for i in range(96):
        np.save(os.path.join(tmp, f"{i:04d}.npy"), synthetic_mel(300 + 7 * (i % 23), 80, i))
    for prec in (sys.argv[1:] or ["fp32", "bf16"]):
        cfg = yaml.safe_load(yaml.safe_dump(base))
        cfg["data"].update(data_dir=tmp, output_dir=None, validation_split=0.0, crop_len=[256, 192, 128], batch_size=8)
        cfg["training"].update(precision=prec, warmup_steps=10, num_epochs=4)
        cfg["logging"]["log_interval"] = 1
        cfg["logging"]["save_interval"] = 1000
        torch.manual_seed(0)
        ops.manual_seed(0)
        logs = []
        tr = Trainer(cfg, device="cuda", log_fn=logs.append)
        for ep in range(1, 5):
            tr._train_epoch(ep)
        torch.cuda.synchronize()
        ok = all(np.isfinite(r[k]) for r in logs for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan"))
        first = np.mean([r["loss_recon_post"] for r in logs[:6]])
        last = np.mean([r["loss_recon_post"] for r in logs[-6:]])
        print(f"{prec}: {len(logs)} steps, finite {ok}, loss_recon_post {first:.4f} -> {last:.4f}, "
              f"loss_d {logs[0]['loss_d']:.4f} -> {logs[-1]['loss_d']:.4f}, {logs[-1]['frames_per_s']:.0f} frames/s", flush=True)
        assert ok and last < first
        del tr
        torch.cuda.empty_cache()
print("SOAK OK")
