"""The CPU oracle (oracle/nn.py) against the golden fixtures produced by running
the reference (tests/golden/gen_golden.py).  fp32, same machine class: 2e-5 abs+rel."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import nn as O

TOL = dict(rtol=2e-5, atol=2e-6)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, **tol)


def mask_from_lens(lens, T):
    return (torch.arange(T)[None, :] >= torch.as_tensor(lens)[:, None]).unsqueeze(1)


def check_grads(P, g, prefix, **kw):
    for k, p in P.items():
        key = prefix + k
        if key in g.files:
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            close(got, g[key], **kw)


def test_fsq_g1():
    g = load_golden("g1_fsq.npz")
    levels = g["levels"].tolist()
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    codes, idx = O.fsq_forward(z, levels)
    assert idx.dtype == torch.int32
    np.testing.assert_array_equal(idx.numpy(), g["idx"])
    np.testing.assert_array_equal(codes.detach().numpy(), g["codes"])
    (codes * torch.from_numpy(g["wgt"])).sum().backward()
    close(z.grad, g["grad_z"])
    dec = O.fsq_indices_to_codes(torch.arange(1000), levels)
    np.testing.assert_array_equal(dec.numpy(), g["decode_all"])
    np.testing.assert_array_equal(dec.numpy(), g["implicit_codebook"])
    # decode is the exact inverse of encode
    np.testing.assert_array_equal(O.fsq_indices_to_codes(idx.long(), levels).numpy(), g["codes"])


def test_convblock2d_g2():
    g = load_golden("g2_convblock2d.npz")
    P = O.params_from_npz(g, "sd/")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    mask = mask_from_lens(g["lens"], x.shape[-1])
    out, hidden = O.convblock2d(P, "", x, mask, return_hidden=True)
    close(out, g["out"])
    close(hidden, g["hidden"])
    (out * torch.from_numpy(g["r"])).sum().backward()
    close(x.grad, g["grad_x"])
    check_grads(P, g, "grad/", atol=2e-5)
    # QUIRK: padded frames equal conv_out.bias
    close(out[1, :, 17:], np.full((16, 7), g["sd/conv_out.bias"][0], np.float32))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_resblock1d_g3(tag):
    g = load_golden("g3_resblock1d.npz")
    cin, cout, k, causal = g[f"{tag}/cfg"].tolist()
    P = O.params_from_npz(g, f"{tag}/sd/")
    x = torch.from_numpy(g[f"{tag}/x"]).requires_grad_(True)
    mask = mask_from_lens(g[f"{tag}/lens"], x.shape[-1])
    out = O.resblock1d(P, "", x, mask, causal=bool(causal))
    close(out, g[f"{tag}/out"])
    (out * torch.from_numpy(g[f"{tag}/r"])).sum().backward()
    close(x.grad, g[f"{tag}/grad_x"], atol=1e-5)
    check_grads(P, g, f"{tag}/grad/", atol=2e-5)


def test_refiner_g4():
    g = load_golden("g4_refiner.npz")
    P = O.params_from_npz(g, "sd/")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    T = x.shape[2]
    mask = torch.arange(T)[None, :] >= torch.from_numpy(g["lens"])[:, None]
    out = O.unet_refiner(P, "", x, mask, depth=3)
    close(out, g["out"])
    (out * torch.from_numpy(g["r"])).sum().backward()
    close(x.grad, g["grad_x"], atol=1e-5)
    check_grads(P, g, "grad/", atol=3e-5)


def _unflat(g, prefix):
    if prefix in g.files:
        return g[prefix]
    n = int(g[prefix + "/len"][0])
    return [_unflat(g, f"{prefix}/{i}") for i in range(n)]


@pytest.mark.parametrize("tag", ["patch", "mbd"])
def test_discriminators_g5(tag):
    g = load_golden("g5_discriminators.npz")
    lens = torch.from_numpy(g["lens"])
    P = O.params_from_npz(g, f"{tag}/sd0/")
    if tag == "patch":
        strides = O.patch_strides([[1, 2], [2, 2], [2, 2], [2, 1], [2, 1], [2, 1]], 6)
        run = lambda x, tr: O.patch_discriminator(P, "", x, lens, strides, tr, True)
    else:
        run = lambda x, tr: O.multibin_discriminator(P, "", x, lens, 8, 2, tr, True)
    for mode, xk, train in (("t", "x1", True), ("e", "x2", False)):
        for p in P.values():
            p.grad = None
        x = torch.from_numpy(g[f"{tag}/{xk}"]).requires_grad_(True)
        logits, pmask, feats = run(x, train)
        ll = [logits] if isinstance(logits, torch.Tensor) else logits
        ml = [pmask] if isinstance(pmask, torch.Tensor) else pmask
        gl = _unflat(g, f"{tag}/{mode}/logits")
        gm = _unflat(g, f"{tag}/{mode}/mask")
        gl = [gl] if isinstance(gl, np.ndarray) else gl
        gm = [gm] if isinstance(gm, np.ndarray) else gm
        for a, b in zip(ll, gl):
            close(a, b, rtol=1e-4, atol=1e-6)
        for a, b in zip(ml, gm):
            np.testing.assert_array_equal(a.numpy(), b)
        gf = _unflat(g, f"{tag}/{mode}/feats")
        ff = [feats] if tag == "patch" else feats
        gf = [gf] if tag == "patch" else gf
        for fa, fb in zip(ff, gf):
            assert len(fa) == len(fb)
            for (t, m), (gt, gmk) in zip(fa, fb):
                close(t, gt, rtol=1e-4, atol=1e-6)
                np.testing.assert_array_equal(m.numpy(), gmk)
        rs = _unflat(g, f"{tag}/t/r")
        sum((l * torch.from_numpy(r)).sum() for l, r in zip(ll, rs)).backward()
        close(x.grad, g[f"{tag}/{mode}/grad_x"], rtol=1e-4, atol=1e-6)
        check_grads(P, g, f"{tag}/{mode}/grad/", rtol=2e-4, atol=1e-5)
        # u / v after the pass (power iteration only in train mode)
        for k, p in P.items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                close(p, g[f"{tag}/sd1/{k}"], rtol=1e-5, atol=1e-6)


def test_lsgan_g6():
    g = load_golden("g6_losses.npz")
    st = O.LSGANState()
    for i in range(9):
        real = torch.from_numpy(g[f"d{i}/real"]).requires_grad_(True)
        fake = torch.from_numpy(g[f"d{i}/fake"]).requires_grad_(True)
        mask = torch.from_numpy(g[f"d{i}/mask"])
        loss = O.lsgan_d_loss(st, real, fake, mask, mask)
        close(loss, g[f"d{i}/loss"])
        if loss.requires_grad:
            loss.backward()
            close(real.grad, g[f"d{i}/grad_real"])
            close(fake.grad, g[f"d{i}/grad_fake"])
        close(st.ema_real, g[f"d{i}/ema_real"])
        close(st.ema_fake, g[f"d{i}/ema_fake"])
    for i in range(3):
        fake = torch.from_numpy(g[f"g{i}/fake"]).requires_grad_(True)
        loss = O.lsgan_g_loss(fake, torch.from_numpy(g[f"g{i}/mask"]))
        close(loss, g[f"g{i}/loss"])
        loss.backward()
        close(fake.grad, g[f"g{i}/grad_fake"])


def test_mel_losses_g6():
    g = load_golden("g6_losses.npz")
    y = torch.from_numpy(g["mel/y"])
    lens = torch.from_numpy(g["mel/lens"])
    for key, kw in (("mel1", dict(group_size=1)), ("mel16", dict(group_size=16)),
                    ("melc", dict(group_size=1, loss_type="charbonnier"))):
        x = torch.from_numpy(g["mel/x"]).requires_grad_(True)
        loss = O.masked_mel_loss(x, y, lens, **kw)
        close(loss, g[f"{key}/loss"])
        loss.backward()
        close(x.grad, g[f"{key}/grad_x"])
    p = torch.from_numpy(g["mae/pred"]).requires_grad_(True)
    loss = O.masked_mae(p, torch.from_numpy(g["mae/target"]), torch.from_numpy(g["mae/mask"]))
    close(loss, g["mae/loss"])
    loss.backward()
    close(p.grad, g["mae/grad_pred"])


def test_encdec_g9():
    g = load_golden("g9_encdec.npz")
    P = O.params_from_npz(g, "sd/", requires_grad=False)
    levels = [8, 5, 5, 5]
    x = torch.from_numpy(g["x"])
    lens = torch.from_numpy(g["lens"])
    mask3 = mask_from_lens(lens, x.shape[1])
    with torch.no_grad():
        idx = O.preencoder_encode(P, x, mask3, levels)
        assert idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.numpy(), g["idx"])
        dec, hid = O.preencoder_decode(P, idx, mask3, levels, return_hidden=True)
        close(dec, g["dec"], atol=1e-5)
        close(hid, g["hid"], atol=1e-5)
        rec, post = O.preencoder_forward(P, x, lens, levels)
        close(rec, g["fwd_recon"], atol=1e-5)
        close(post, g["fwd_post"], atol=1e-5)
        np.testing.assert_array_equal(O.preencoder_encode(P, x, None, levels).numpy(), g["idx_nomask"])
        close(O.preencoder_decode(P, torch.from_numpy(g["idx_nomask"]), None, levels), g["dec_nomask"], atol=1e-5)


def test_init_matches_manifest_g8():
    """The oracle's random-init helper produces exactly the reference's state_dict
    keys and shapes for both shipped architectures."""
    import yaml
    from oracle import train_step as TS
    man = json.load(open(os.path.join(GOLDEN, "g8_manifest.json")))
    arch = {"hifispeech": dict(channels=[512, 512, 512, 768], kernel_sizes=[3, 3, 5, 7], fsq_levels=[8, 5, 5, 5],
                               refiner_base_channels=64, refiner_depth=3),
            "hifimusic": dict(channels=[384, 384, 512, 512], kernel_sizes=[3, 3, 5, 7], fsq_levels=[8, 5, 5, 5],
                              refiner_base_channels=96, refiner_depth=3)}
    for name, mel in (("hifispeech", 80), ("hifimusic", 128)):
        cfg = {"model": {"mel_channels": mel, "generator": arch[name]}}
        P = TS.init_generator_params(cfg)
        want = man[f"{name}@{mel}"]["generator"]
        assert set(P) == set(want)
        for k, v in P.items():
            assert list(v.shape) == want[k][0], k
