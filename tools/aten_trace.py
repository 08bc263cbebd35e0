"""Which ATen kernels (adds, fills, copies) a training step still launches, grouped by operator, shape and the
innermost mqgan_amd frame that issued them.   python tools/aten_trace.py hifispeech fp32"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

ge.build()
import bench
from mqgan_amd.train import Trainer

cfg_name, precision = sys.argv[1], sys.argv[2]
cfg = bench.load_config(cfg_name)
cfg["training"]["precision"] = precision
dev = torch.device("cuda:0")
B, T = cfg["data"]["batch_size"], cfg["data"]["crop_len"]
mel = cfg["model"]["mel_channels"]
tr = Trainer(cfg, device=dev, build_data=False)
tr.generator.train(); tr.patch_discriminator.train(); tr.multibin_discriminator.train()
tr.epoch = 1
batches = bench.synthetic_batches(B, T, mel, 1, dev, 0)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(2):
    tr.train_step(batches[0], lens)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=True) as prof:
    tr.train_step(batches[0], lens)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    self_dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if self_dt <= 0:
        continue
    frame = "?"
    for fr in (ev.stack or []):
        if "mqgan_amd" in fr or "bench.py" in fr:
            frame = fr.split("/")[-1]
            break
    if frame == "?":                   # launched by the autograd engine: name the enclosing node instead
        par, chain = ev.cpu_parent, []
        while par is not None and len(chain) < 3:
            chain.append(par.name.replace("autograd::engine::evaluate_function: ", "eval:"))
            par = par.cpu_parent
        frame = " < ".join(chain) if chain else "?"
    key = (ev.name, '', frame[:110])
    agg[key][0] += 1
    agg[key][1] += self_dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"ATen device time in one step: {tot / 1e3:.2f} ms over {sum(v[0] for _, v in rows)} ops")
for (name, shp, frame), (n, t) in rows[:45]:
    print(f"{t / 1e3:8.3f} ms {n:5d}x  {name:14s} {frame}")
