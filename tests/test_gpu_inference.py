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
    export(str(ckpt), str(cfg), str(out))
    assert (out / "model_config.yaml").read_text() == cfg.read_text()
    return str(out), str(ckpt), str(cfg)


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
