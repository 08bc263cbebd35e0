"""-m gpu: the inference drop-ins: export -> model directory -> ScriptedPreEncoder ->
reencode_spectrograms, against the golden encode/decode fixture G9 and the reference's error behaviour."""
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden

pytestmark = pytest.mark.gpu

from gpu_util import DEV, close, npz_state_dict, t  # noqa: E402
from test_oracle_trainstep import TINY_CFG  # noqa: E402


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    import __graft_entry__ as g
    g.build()
    from mqgan_amd.export import export
    d = tmp_path_factory.mktemp("export")
    g9 = load_golden("g9_encdec.npz")
    ckpt = d / "checkpoint_epoch_001.pth"
    torch.save({"epoch": 1, "model_state_dict": {"module." + k: v for k, v in npz_state_dict(g9, "sd/").items()}}, ckpt)
    cfg = d / "cfg.yaml"
    cfg.write_text(yaml.safe_dump({"model": TINY_CFG["model"]}))
    out = d / "model"
    export(str(ckpt), str(cfg), str(out))                         # default payload: folded weights
    assert (out / "model_config.yaml").read_text() == cfg.read_text()
    export(str(ckpt), str(cfg), str(d / "model_unfolded"), folded=False)
    return str(out), str(ckpt), str(cfg), str(d / "model_unfolded")


def test_scripted_preencoder_matches_reference_fixture(model_dir):
    from mqgan_amd.scripted_preencoder import ScriptedPreEncoder
    g = load_golden("g9_encdec.npz")
    m = ScriptedPreEncoder(model_dir[0], device="cuda")
    assert m.mel_channels == 16 and m.fsq_levels == [8, 5, 5, 5]
    x = torch.from_numpy(g["x"])                       # CPU input is moved by the wrapper, as in the reference
    lens = g["lens"].tolist()
    idx = m.encode(x, lengths=lens)
    assert idx.dtype == torch.int64
    np.testing.assert_array_equal(idx.cpu().numpy(), g["idx"])
    close(m.decode(idx, lengths=lens), g["dec"], rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(m.encode(x).cpu().numpy(), g["idx_nomask"])
    # frozen effective weights: a second call gives identical results
    np.testing.assert_array_equal(m.encode(x, lengths=lens).cpu().numpy(), g["idx"])
    with pytest.raises(ValueError):
        m.encode(x[0], lengths=[40])
    with pytest.raises(FileNotFoundError):
        ScriptedPreEncoder(model_dir[0] + "_missing")


def test_reencode_tree_and_bad_files(model_dir, tmp_path):
    from mqgan_amd.reencode_spectrograms import reencode_spectrograms, reencode_spectrograms_from_checkpoint
    src, dst1, dst2 = tmp_path / "in", tmp_path / "out1", tmp_path / "out2"
    (src / "a" / "b").mkdir(parents=True)
    rng = np.random.default_rng(0)
    lens = [37, 64, 5, 50, 41]
    for i, L in enumerate(lens):
        np.save(src / ("a" if i % 2 else "a/b") / f"m{i}.npy", (rng.standard_normal((L, 16)) * 2 - 4).astype(np.float32))
    np.save(src / "bad.npy", np.zeros((7,), np.float32))           # wrong rank: its batch is skipped, the rest goes on
    n1 = reencode_spectrograms(model_dir[0], str(src), str(dst1), "cuda", 2)
    cfg = yaml.safe_load(open(model_dir[2]))
    n2 = reencode_spectrograms_from_checkpoint(model_dir[1], cfg, str(src), str(dst2), "cuda", 2)
    assert n1 == n2 and 3 <= n1 <= 5
    found = 0
    for root, _, files in os.walk(dst1):
        for f in files:
            a = np.load(os.path.join(root, f))
            rel = os.path.relpath(os.path.join(root, f), dst1)
            b = np.load(os.path.join(dst2, rel))
            orig = np.load(os.path.join(src, rel))
            assert a.shape == orig.shape and a.dtype == np.float32
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)    # folded vs unfolded weight norm agree
            assert np.isfinite(a).all()
            found += 1
    assert found == n1


def test_folded_and_unfolded_payloads_agree(model_dir):
    """Export formats: the folded payload holds plain effective weights only - no weight-norm keys of
    either flavour -, is smaller, and ScriptedPreEncoder gives the same indices and mels from both."""
    from mqgan_amd.scripted_preencoder import PAYLOAD, ScriptedPreEncoder
    g = load_golden("g9_encdec.npz")
    pf = torch.load(os.path.join(model_dir[0], PAYLOAD), map_location="cpu", weights_only=False)
    pu = torch.load(os.path.join(model_dir[3], PAYLOAD), map_location="cpu", weights_only=False)
    assert pf["folded"] and pf["format"] == "mqgan_amd/2-folded" and not pu["folded"]
    assert not any(("original" in k) or k.endswith("weight_g") or k.endswith("weight_v") for k in pf["model_state_dict"])
    assert any("parametrizations.weight.original0" in k for k in pu["model_state_dict"])
    assert set(pu["model_state_dict"]) == set(npz_state_dict(g, "sd/"))          # the reference's training key layout
    assert sum(v.numel() for v in pf["model_state_dict"].values()) < sum(v.numel() for v in pu["model_state_dict"].values())
    a, b = ScriptedPreEncoder(model_dir[0], device="cuda"), ScriptedPreEncoder(model_dir[3], device="cuda")
    x, lens = torch.from_numpy(g["x"]), g["lens"].tolist()
    ia, ib = a.encode(x, lengths=lens), b.encode(x, lengths=lens)
    np.testing.assert_array_equal(ia.cpu().numpy(), g["idx"])
    np.testing.assert_array_equal(ib.cpu().numpy(), g["idx"])
    close(a.decode(ia, lengths=lens), b.decode(ib, lengths=lens), rtol=1e-5, atol=1e-6)
    close(a.decode(ia, lengths=lens), g["dec"], rtol=1e-4, atol=2e-5)


def test_length_bucketed_reencode(model_dir, tmp_path):
    """SURVEY 8f N2.  'exact' buckets (identical lengths only, no padding): every file gets bit-for-bit the result of being
    processed alone, whatever the directory order.  'sorted' buckets (length-sorted neighbours under a padding budget):
    every file is written with its own length, the longest file of each batch still equals its stand-alone result bit
    for bit, the padded ones agree with it on frames away from their end (the reference's padded frames leak into the
    last valid ones), and much less padding is processed than in directory order."""
    from mqgan_amd.reencode_spectrograms import _find_npy, _length_of, plan_batches, reencode_spectrograms
    src = tmp_path / "in"
    (src / "x" / "y").mkdir(parents=True)
    rng = np.random.default_rng(1)
    lens = [40, 12, 33, 40, 12, 64, 33, 12, 57, 40, 5, 64]
    for i, L in enumerate(lens):
        np.save(src / ("x" if i % 3 == 0 else ("x/y" if i % 3 == 1 else "")) / f"u{i:02d}.npy",
                (rng.standard_normal((L, 16)) * 2 - 4).astype(np.float32))
    outs, stats = {}, {}
    for name, kw in (("alone", dict(batch_size=1, bucket="order")), ("exact", dict(batch_size=4, bucket="exact")),
                     ("sorted", dict(batch_size=4, bucket="sorted", max_pad=0.25)), ("order", dict(batch_size=4, bucket="order"))):
        dst = tmp_path / name
        stats[name] = {}
        n = reencode_spectrograms(model_dir[0], str(src), str(dst), "cuda", stats=stats[name], **kw)
        assert n == len(lens)
        outs[name] = {os.path.relpath(p, dst): np.load(p) for p in _find_npy(str(dst))}
        assert len(outs[name]) == len(lens)
    for rel, a in outs["alone"].items():
        L = np.load(src / rel).shape[0]
        assert a.shape == (L, 16)
        np.testing.assert_array_equal(outs["exact"][rel], a)                       # no padding -> identical bits
        b = outs["sorted"][rel]
        assert b.shape == a.shape
        if L > 16:
            np.testing.assert_allclose(b[:L - 12], a[:L - 12], rtol=2e-3, atol=2e-3)
    files = _find_npy(str(src))
    plan = plan_batches(files, [_length_of(p) for p in files], 4, "sorted", 0.25)
    for batch in plan:                                                             # the longest member of a batch is unpadded
        rel = os.path.relpath(max(batch, key=_length_of), src)
        np.testing.assert_array_equal(outs["sorted"][rel], outs["alone"][rel])
    assert stats["exact"]["pad_fraction"] == 0.0
    assert stats["sorted"]["pad_fraction"] <= 0.25 + 1e-9 and stats["order"]["pad_fraction"] > 0.0
    assert stats["sorted"]["frames"] == sum(lens) and stats["sorted"]["frames_per_s"] > 0
