#!/usr/bin/env python3
"""Where do the milliseconds go when the config-2 step is fed by the real input pipeline (bench.py: config2_through_the_loader)?
Per step of Trainer._train_epoch over synthetic .npy files: host time blocked waiting for the next batch, host time to issue the
step, GPU time between step boundaries (HIP events), plus the epoch's start-up and drain.  usage: python tools/loader_probe.py [epochs]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import bench  # noqa: E402
from mqgan_amd import data as D  # noqa: E402
from mqgan_amd.train import Trainer, synthetic_mel  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = bench.load_config("hifispeech")
B, T, mel = cfg["data"]["batch_size"], cfg["data"]["crop_len"], cfg["model"]["mel_channels"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with tempfile.TemporaryDirectory() as tmp:
    for i in range(576):
        np.save(os.path.join(tmp, f"{i:05d}.npy"), synthetic_mel(640, mel, 50000 + i))
    cfg["data"]["data_dir"], cfg["data"]["output_dir"], cfg["data"]["validation_split"] = tmp, None, 0.0
    tr = Trainer(cfg, device=dev, build_data=True)
    # instrument: time blocked in the queue, and the step itself
    waits, issues, marks = [], [], []
    real_step = tr.train_step

    def step(real, lens):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        t0 = time.perf_counter()
        out = real_step(real, lens)
        issues.append((time.perf_counter() - t0) * 1e3)
        return out

    tr.train_step = step
    orig_iter = D.PrefetchLoader.epoch

    def timed_iter(self, epoch, next_epoch=None):
        it = orig_iter(self, epoch, next_epoch)
        while True:
            t0 = time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                return
            waits.append((time.perf_counter() - t0) * 1e3)
            yield item

    D.PrefetchLoader.epoch = timed_iter
    for ep in range(epochs):
        waits.clear(); issues.clear(); marks.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr._train_epoch(ep + 1, next_epoch=ep + 2 if ep + 1 < epochs else None)
        t_host = (time.perf_counter() - t0) * 1e3
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) * 1e3
        gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)] + [marks[-1].elapsed_time(end)]
        n = len(gpu)
        print(f"epoch {ep}: {n} steps, wall {t_all:.1f} ms = {t_all / n:.1f} ms/step; host loop returned after {t_host:.1f} ms")
        print("  host wait for batch (ms): " + " ".join(f"{w:.0f}" for w in waits))
        print("  host issue of step  (ms): " + " ".join(f"{w:.0f}" for w in issues))
        print("  GPU step-to-step    (ms): " + " ".join(f"{w:.0f}" for w in gpu))
        print(f"  sums: wait {sum(waits):.0f}, issue {sum(issues):.0f}, GPU {sum(gpu):.0f}; median GPU step {sorted(gpu)[n // 2]:.1f}")
