"""Element volume through the pointwise kernels in one training step, by activation kind and storage type.
  python tools/pw_volume.py hifimusic bf16"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

ge.build()
import bench
from mqgan_amd import ops, _lib

cfg_name, precision = sys.argv[1], sys.argv[2]
lib = _lib.load()
VERBOSE = False
vol = collections.Counter()
cnt = collections.Counter()


def wrap(name):
    orig = getattr(lib, name)

    def f(d, *a):
        o = d._obj
        key = (name, o.act, bool(o.p_drop > 0))
        vol[key] += o.rows * o.C
        cnt[key] += 1
        if VERBOSE and o.act == 0:
            print("   ", name, "rows", o.rows, "C", o.C, "mask_post", o.mask_post, flush=True)
        return orig(d, *a)
    setattr(lib, name, f)


for nm in ("mq_pointwise_fwd", "mq_pointwise_bwd", "mq_pointwise_fwd_h", "mq_pointwise_bwd_h"):
    wrap(nm)
from mqgan_amd.train import Trainer
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
vol.clear(); cnt.clear()
VERBOSE = len(sys.argv) > 3
tr.train_step(batches[0], lens)
torch.cuda.synchronize()
for k in sorted(vol, key=lambda k: -vol[k]):
    print(f"{k[0]:22s} act={k[1]} drop={k[2]!s:5s} calls {cnt[k]:4d}  {vol[k] / 1e9:7.3f} G elements")
