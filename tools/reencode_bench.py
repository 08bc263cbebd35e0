#!/usr/bin/env python3
"""Inference throughput of the bulk re-encoding job on synthetic spectrograms: hifispeech PreEncoder
(random weights), N files with lengths drawn from U(200, 1200), batch 32, for the three batching policies of
mqgan_amd.reencode_spectrograms.  Prints valid mel-frames/s and the share of padding frames that were processed.
usage: python tools/reencode_bench.py [n_files]"""
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
from mqgan_amd.export import export  # noqa: E402
from mqgan_amd.preencoder import PreEncoder  # noqa: E402
from mqgan_amd.reencode_spectrograms import reencode_spectrograms  # noqa: E402
from mqgan_amd.train import synthetic_mel  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with open(os.path.join(ROOT, "configs", "model_config_hifispeech.yaml")) as f:
    cfg = yaml.safe_load(f)
m, g = cfg["model"], cfg["model"]["generator"]
with tempfile.TemporaryDirectory() as tmp:
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
    for bucket in ("sorted", "sorted", "exact", "order"):          # the first pass warms allocator and kernels up
        st = {}
        reencode_spectrograms(mdir, src, os.path.join(tmp, "out_" + bucket), "cuda", 32, bucket=bucket, stats=st)
        print(f"{bucket:7s} {st['frames_per_s']:10.0f} mel-frames/s   {st['files']} files, {st['frames']} frames, "
              f"{100 * st['pad_fraction']:5.1f} % padding, {st['seconds']:.2f} s", flush=True)
