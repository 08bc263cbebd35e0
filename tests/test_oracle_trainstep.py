"""oracle.train_step against fixtures G7 / G7b: two optimisation steps driven by the
reference's own Trainer._train_discriminator / _train_generator (train.py:380-501)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import nn as O
from oracle import train_step as TS

TINY_CFG = {
    "model": {"mel_channels": 16,
              "generator": {"channels": [16, 16, 24], "kernel_sizes": [3, 5], "fsq_levels": [8, 5, 5, 5],
                            "dropout": 0.1, "refiner_base_channels": 8, "refiner_depth": 3},
              "discriminator_patch": {"hidden_channels": [8, 8, 12, 16, 16], "kernel_sizes": [5, 5, 5, 3, 3, 3],
                                      "strides": [[1, 2], [2, 2], [2, 2], [2, 1], [2, 1], [2, 1]]},
              "discriminator_multibin": {"hidden_channels": [8, 8, 16, 16, 24], "kernel_sizes": [7, 5, 3, 3, 3, 3],
                                         "n_bins": 8, "n_no_strides": 2}},
    "training": {"lr": 1e-4, "beta1": 0.9, "beta2": 0.999, "lr_d_factor": 1.15, "d_beta1": 0.5, "d_beta2": 0.999,
                 "warmup_steps": 1000, "discriminator_train_start_epoch": 0,
                 "loss_weights": {"fm_lambda": 0.25, "Gloss_lambda": 15.0, "recon_lambda": 15.0},
                 "use_fm_loss": False, "seed": 42},
}


@pytest.mark.parametrize("fname,use_fm", [("g7_trainstep.npz", False), ("g7b_trainstep_fm.npz", True)])
def test_two_train_steps(fname, use_fm):
    g = load_golden(fname)
    cfg = {"model": TINY_CFG["model"], "training": dict(TINY_CFG["training"], use_fm_loss=use_fm)}
    Pg = O.params_from_npz(g, "g0/")
    Pp = O.params_from_npz(g, "p0/")
    Pm = O.params_from_npz(g, "m0/")
    tr = TS.OracleTrainer(Pg, Pp, Pm, cfg, gen_training=False)
    lens = torch.from_numpy(g["lens"])
    for s in range(2):
        out = tr.step(torch.from_numpy(g[f"s{s}/x"]), lens)
        np.testing.assert_allclose(out["recon_pre"].numpy(), g[f"s{s}/recon_pre"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out["recon_post"].numpy(), g[f"s{s}/recon_post"], rtol=1e-4, atol=1e-5)
        for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan", "loss_fm"):
            np.testing.assert_allclose(out[k], float(g[f"s{s}/{k}"]), rtol=1e-4, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(out["learning_rate"], float(g[f"s{s}/lr"]), rtol=1e-12)
        np.testing.assert_allclose(tr.last_norm_d, float(g[f"s{s}/preclip_norm_d"]), rtol=1e-4)
        np.testing.assert_allclose(tr.last_norm_g, float(g[f"s{s}/preclip_norm_g"]), rtol=1e-4)
        np.testing.assert_allclose([tr.gan.ema_real.item(), tr.gan.ema_fake.item()], g[f"s{s}/ema"], rtol=1e-4, atol=1e-7)
        for tag, P in (("gsum", Pg), ("psum", Pp), ("msum", Pm)):
            for k, v in P.items():
                want = g[f"s{s}/{tag}/{k}"]
                v = v.detach().double()
                np.testing.assert_allclose([v.sum().item(), v.norm().item()], want, rtol=1e-5, atol=1e-6,
                                           err_msg=f"{tag}/{k}")
    for k in [f for f in g.files if f.startswith("gfinal/")]:
        np.testing.assert_allclose(Pg[k[7:]].detach().numpy(), g[k], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(Pp["convs.1.weight_orig"].detach().numpy(), g["pfinal/convs.1.weight_orig"], rtol=1e-5,
                               atol=1e-7)
    np.testing.assert_allclose(Pp["convs.1.weight_u"].numpy(), g["pfinal/convs.1.weight_u"], rtol=1e-5, atol=1e-7)
