"""ScriptedPreEncoder with the reference's class API and model-directory convention (reference
scripted_preencoder.py:24-175), backed by the HIP PreEncoder instead of a TorchScript trace.

Model directory = ``model_config.yaml`` (the training YAML, copied verbatim as the reference's exporter
does, convert_to_torchscript.py:94-97) + ``model_hip.pt`` (a ``torch.save`` of ``{"model_state_dict": ..., "folded": bool}``
written by ``mqgan_amd.export``: folded inference weights by default, or the reference's training key layout).  The reference's ``model_cpu.pt`` /
``model_cuda.pt`` are TorchScript traces of the reference's own Python and are not read.
"""
import os
from typing import List, Optional, Union

import torch
import yaml

from .preencoder import PreEncoder, sequence_mask

PAYLOAD = "model_hip.pt"


def build_from_config(config: dict) -> PreEncoder:
    m = config["model"]
    g = m["generator"]
    return PreEncoder(mel_channels=m["mel_channels"], channels=g["channels"], kernel_sizes=g["kernel_sizes"], dropout=0.0,
                      fsq_levels=g["fsq_levels"], refiner_base_channels=g.get("refiner_base_channels", 128),
                      refiner_depth=g.get("refiner_depth", 3),
                      refiner_hidden_proj_divisor=g.get("refiner_hidden_proj_divisor", 8))


def freeze_weights(model: torch.nn.Module):
    from .nn_base import ConvParams
    for mod in model.modules():
        if isinstance(mod, ConvParams):
            mod.freeze_()


class ScriptedPreEncoder:
    def __init__(self, model_dir: str, device: Optional[str] = "cuda"):
        if not os.path.isdir(model_dir):
            raise FileNotFoundError(f"Model directory not found: {model_dir}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            # the reference defaults to device='cpu' and falls back cuda -> cpu silently (scripted_preencoder.py:38-94); this
            # package has no CPU path, so it says so instead of pretending
            raise RuntimeError(f"mqgan_amd.ScriptedPreEncoder runs on the HIP device only, not on device={device!r}: pass device='cuda' "
                               "(the default here; `reencode_spectrograms.py --device cuda`).  There is no CPU path - the reference's "
                               "CPU default has no counterpart (INTEGRATION.md, section 1).")
        if not torch.cuda.is_available():
            raise RuntimeError("mqgan_amd.ScriptedPreEncoder needs an MI355X (torch.cuda.is_available() is False); unlike the reference "
                               "it does not fall back to the CPU")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)               # kernels launch on the current device's stream
        config_path = os.path.join(model_dir, "model_config.yaml")
        if not os.path.exists(config_path):
            raise FileNotFoundError(f"model_config.yaml not found in: {model_dir}")
        try:
            with open(config_path, "r") as f:
                self.config = yaml.safe_load(f)
        except Exception as e:
            raise RuntimeError(f"Failed to load or parse config file: {e}")
        model_path = os.path.join(model_dir, PAYLOAD)
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"{PAYLOAD} not found in: {model_dir}")
        try:
            payload = torch.load(model_path, map_location="cpu", weights_only=False)
            self.model = build_from_config(self.config)
            if payload.get("folded"):                  # format mqgan_amd/2-folded: plain effective weights everywhere
                from .nn_base import ConvParams
                for mod in self.model.modules():
                    if isinstance(mod, ConvParams):
                        mod.fold_weight_norm_(both=True)
            self.model.load_state_dict(payload["model_state_dict"], strict=True)
            self.model.to(self.device).eval()
            freeze_weights(self.model)
        except Exception as e:
            raise RuntimeError(f"Failed to load model from {model_path}: {e}")

    @property
    def mel_channels(self) -> int:
        return self.config.get("model", {}).get("mel_channels", 0)

    @property
    def fsq_levels(self) -> List[int]:
        return self.config.get("model", {}).get("generator", {}).get("fsq_levels", [])

    def _prepare_mask(self, max_len: int, lengths: torch.Tensor) -> torch.Tensor:
        return sequence_mask(max_len, lengths.to(self.device)).unsqueeze(1)          # (B, 1, T), True = padded

    def encode(self, spectrogram: torch.Tensor, lengths: Optional[Union[List[int], torch.Tensor]] = None) -> torch.Tensor:
        if spectrogram.ndim != 3:
            raise ValueError(f"Input spectrogram must be a 3D tensor (B, T, C), but got shape {spectrogram.shape}")
        spectrogram = spectrogram.to(self.device).float().contiguous()
        mask = None
        if lengths is not None:
            if not isinstance(lengths, torch.Tensor):
                lengths = torch.tensor(lengths, dtype=torch.long)
            mask = self._prepare_mask(spectrogram.shape[1], lengths)
        with torch.no_grad():
            try:
                return self.model.encode(spectrogram, mask)
            except Exception as e:
                raise RuntimeError(f"An error occurred during the encode operation: {e}")

    def decode(self, indices: torch.Tensor, lengths: Optional[Union[List[int], torch.Tensor]] = None) -> torch.Tensor:
        indices = indices.to(self.device)
        mask = None
        if lengths is not None:
            if not isinstance(lengths, torch.Tensor):
                lengths = torch.tensor(lengths, dtype=torch.long)
            mask = self._prepare_mask(indices.shape[1], lengths)
        with torch.no_grad():
            try:
                return self.model.decode(indices, mask)
            except Exception as e:
                raise RuntimeError(f"An error occurred during the decode operation: {e}")
