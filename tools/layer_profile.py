#!/usr/bin/env python3
"""Per-launch conv timing of one train step (HIP events), grouped by shape: `layer_profile.py [config] [precision]`
(hifispeech fp32 = BASELINE config 2 by default; `hifimusic bf16` = config 3)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mqgan_amd import ops
from mqgan_amd.train import Trainer

cfg = bench.load_config(sys.argv[1] if len(sys.argv) > 1 else "hifispeech")
if len(sys.argv) > 2:
    cfg["training"]["precision"] = sys.argv[2]
dev = torch.device("cuda", 0)
tr = Trainer(cfg, device=dev, build_data=False)
tr.generator.train(); tr.patch_discriminator.train(); tr.multibin_discriminator.train(); tr.epoch = 1
B, T, mel = cfg["data"]["batch_size"], cfg["data"]["crop_len"], cfg["model"]["mel_channels"]
x = bench.synthetic_batches(B, T, mel, 1, dev, 0)[0]
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
tr.train_step(x, lens)
timer = ops.KernelTimer(); ops.TIMER = timer
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); tr.train_step(x, lens); e.record(); torch.cuda.synchronize()
ops.TIMER = None
print(f"step {s.elapsed_time(e):.1f} ms")
agg = collections.OrderedDict()
for (tag, flops, a, b), det in zip(timer.records, timer.details):
    k = (tag.split('<')[0].replace('conv_', '') + ('/dgrad' if ', true,' in tag else ''), det)
    d = agg.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += a.elapsed_time(b); d[2] += flops
names = "N H W C K R S P Q sh sw ph pw dh dw g".split()
tot = 0
for (kind, det), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    dd = dict(zip(names, det))
    print(f"{kind:16s} x{n:<2d} {ms:8.2f} ms {fl / ms / 1e9:7.1f} TF  N{dd['N']} {dd['H']}x{dd['W']} C{dd['C']}->K{dd['K']} k{dd['R']}x{dd['S']} s{dd['sh']},{dd['sw']} g{dd['g']}")
print("total conv ms", tot)
# LAYER_FILTER="k3x5 s1,1": every launch of the matching layers, in launch order, with the kernel instantiation
flt = os.environ.get("LAYER_FILTER")
if flt:
    for (tag, flops, a, b), det in zip(timer.records, timer.details):
        dd = dict(zip(names, det))
        shape = f"N{dd['N']} {dd['H']}x{dd['W']} C{dd['C']}->K{dd['K']} k{dd['R']}x{dd['S']} s{dd['sh']},{dd['sw']} g{dd['g']}"
        if flt in shape:
            ms = a.elapsed_time(b)
            print(f"  {shape:52s} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF  {tag}")
