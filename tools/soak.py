import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, yaml
from mqgan_amd.train import Trainer, synthetic_mel
from mqgan_amd import ops
# usage: soak.py <precision | config:precision> ...   (config: hifispeech (default) or hifimusic - the 96-channel refiner, i.e. the
# kernels of igemm_h2.hip that only that architecture reaches)
for arg in sys.argv[1:]:
    name, prec = arg.split(":") if ":" in arg else ("hifispeech", arg)
    cfg = yaml.safe_load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), f"configs/model_config_{name}.yaml")))
    cfg["training"]["warmup_steps"] = 20
    B, T, mel = 8, 256, cfg["model"]["mel_channels"]
    x = torch.from_numpy(np.stack([synthetic_mel(T, mel, i) for i in range(B)])).cuda()
    lens = torch.tensor([256, 200, 256, 131, 256, 256, 77, 256], dtype=torch.int32).cuda()
    cfg["training"]["precision"] = prec
    torch.manual_seed(0); ops.manual_seed(0)
    tr = Trainer(cfg, device="cuda", build_data=False)
    tr.generator.train(); tr.patch_discriminator.train(); tr.multibin_discriminator.train(); tr.epoch = 1
    hist = []
    for s in range(120):
        out = tr.train_step(x, lens)
        if s % 20 == 0 or s == 119:
            hist.append((s, round(float(out["loss_recon_post"]), 4), round(float(out["loss_recon_pre"]), 4), round(float(out["loss_d"]), 4), round(float(out["loss_gan"]), 4)))
    print(name, prec, hist, flush=True)
