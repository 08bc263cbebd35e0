"""Exporter: training checkpoint + YAML -> model directory for ScriptedPreEncoder (the role of the
reference's convert_to_torchscript.py:42-122, same CLI).  Writes ``model_config.yaml`` (verbatim copy)
and ``model_hip.pt`` (state_dict in the reference's key layout; the legacy weight-norm convs of the
decoder stay unfolded so the file still loads with strict=True), then runs the reference's post-export
self-check: encode -> decode at a length different from anything seen before returns the input's shape."""
import argparse
import os

import torch
import yaml

from .preencoder import get_pre_encoder
from .scripted_preencoder import PAYLOAD, ScriptedPreEncoder


def export(checkpoint: str, config_path: str, output_dir: str, verify: bool = True):
    with open(config_path, "r") as f:
        config_str = f.read()
    config = yaml.safe_load(config_str)
    m, g = config["model"], config["model"]["generator"]
    os.makedirs(output_dir, exist_ok=True)
    model = get_pre_encoder(model_path=checkpoint, device="cpu", mel_channels=m["mel_channels"], channels=g["channels"],
                            kernel_sizes=g["kernel_sizes"], fsq_levels=g["fsq_levels"],
                            refiner_base_channels=g.get("refiner_base_channels", 128), refiner_depth=g.get("refiner_depth", 3),
                            refiner_hidden_proj_divisor=g.get("refiner_hidden_proj_divisor", 8), inference=False)
    torch.save({"model_state_dict": model.state_dict(), "format": "mqgan_amd/1"}, os.path.join(output_dir, PAYLOAD))
    with open(os.path.join(output_dir, "model_config.yaml"), "w") as f:
        f.write(config_str)
    if verify and torch.cuda.is_available():
        wrapper = ScriptedPreEncoder(output_dir, device="cuda")
        eval_len = 384
        spec = torch.randn(2, eval_len, wrapper.mel_channels, device="cuda")
        lengths = torch.tensor([eval_len, eval_len], device="cuda", dtype=torch.long)
        rec = wrapper.decode(wrapper.encode(spec, lengths=lengths), lengths=lengths)
        assert rec.shape == spec.shape, f"Shape mismatch after decode. Expected {spec.shape}, got {rec.shape}"
    return output_dir


def main(argv=None):
    ap = argparse.ArgumentParser(description="Export a PreEncoder checkpoint to a ScriptedPreEncoder model directory.")
    ap.add_argument("--checkpoint", type=str, required=True)
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--output_dir", type=str, required=True)
    a = ap.parse_args(argv)
    export(a.checkpoint, a.config, a.output_dir)
    print(f"exported to {a.output_dir}")


if __name__ == "__main__":
    main()
