"""Batch re-encoding of .npy spectrogram trees (reference reencode_spectrograms.py:8-108 and
reencode_spectrograms_from_checkpoint.py:9-108, same CLIs): load -> zero-pad to the batch maximum ->
encode -> decode -> trim -> save to the mirrored path; a failing batch is reported and skipped."""
import argparse
import os

import numpy as np
import torch
import yaml

from .preencoder import get_pre_encoder, sequence_mask
from .scripted_preencoder import ScriptedPreEncoder, freeze_weights


def _find_npy(input_dir):
    return [os.path.join(root, f) for root, _, files in os.walk(input_dir) for f in files if f.endswith(".npy")]


def _run(encode_decode, input_dir, output_dir, batch_size):
    files = _find_npy(input_dir)
    if not files:
        print("Warning: No .npy files were found.")
        return 0
    done = 0
    for i in range(0, len(files), batch_size):
        paths = files[i:i + batch_size]
        try:
            specs = [np.load(p) for p in paths]
            lens = [s.shape[0] for s in specs]
            max_len = max(lens)
            batch = np.stack([np.concatenate([s, np.zeros((max_len - s.shape[0], s.shape[1]), dtype=s.dtype)], axis=0)
                              for s in specs])
            out = encode_decode(torch.from_numpy(batch).float(), lens)
            for j, p in enumerate(paths):
                rel = os.path.relpath(p, input_dir)
                dst = os.path.join(output_dir, rel)
                os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
                np.save(dst, out[j, :lens[j], :].cpu().numpy())
                done += 1
        except Exception as e:           # reference: print and continue with the next batch (:83-85)
            print(f"\nCould not process batch starting with {paths[0]}. Error: {e}")
            continue
    return done


def reencode_spectrograms(model_path, input_dir, output_dir, device, batch_size):
    """ScriptedPreEncoder variant (model directory)."""
    try:
        model = ScriptedPreEncoder(model_path, device=device)
    except (FileNotFoundError, RuntimeError) as e:
        print(f"Error: Could not load the model. {e}")
        return 0
    return _run(lambda x, lens: model.decode(model.encode(x, lengths=lens), lengths=lens), input_dir, output_dir, batch_size)


def reencode_spectrograms_from_checkpoint(checkpoint_path, config, input_dir, output_dir, device, batch_size):
    """Raw-checkpoint variant (get_pre_encoder(..., inference=True))."""
    try:
        m, g = config["model"], config["model"]["generator"]
        model = get_pre_encoder(model_path=checkpoint_path, device=device, mel_channels=m["mel_channels"], channels=g["channels"],
                                kernel_sizes=g["kernel_sizes"], fsq_levels=g["fsq_levels"],
                                refiner_base_channels=g.get("refiner_base_channels", 128), refiner_depth=g.get("refiner_depth", 3),
                                refiner_hidden_proj_divisor=g.get("refiner_hidden_proj_divisor", 8), inference=True)
        freeze_weights(model)
    except (FileNotFoundError, RuntimeError, KeyError) as e:
        print(f"Error: Could not load the model. {e}")
        return 0

    def enc_dec(x, lens):
        x = x.to(device)
        mask = sequence_mask(x.shape[1], torch.tensor(lens, dtype=torch.long, device=device)).unsqueeze(1)
        with torch.no_grad():
            return model.decode(model.encode(x, x_mask=mask), x_mask=mask)
    return _run(enc_dec, input_dir, output_dir, batch_size)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Re-encode spectrograms with a PreEncoder on MI355X.")
    ap.add_argument("--model", type=str, default=None, help="model directory written by mqgan_amd.export")
    ap.add_argument("--checkpoint", type=str, default=None, help="raw training checkpoint (.pth) - needs --config")
    ap.add_argument("--config", type=str, default=None)
    ap.add_argument("--input_dir", type=str, required=True)
    ap.add_argument("--output_dir", type=str, required=True)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--batch_size", type=int, default=32)
    a = ap.parse_args(argv)
    if a.model:
        n = reencode_spectrograms(a.model, a.input_dir, a.output_dir, a.device, a.batch_size)
    elif a.checkpoint and a.config:
        with open(a.config) as f:
            n = reencode_spectrograms_from_checkpoint(a.checkpoint, yaml.safe_load(f), a.input_dir, a.output_dir, a.device,
                                                      a.batch_size)
    else:
        raise SystemExit("give --model DIR or --checkpoint FILE --config YAML")
    print(f"\nProcessing complete: {n} file(s) written to {a.output_dir}")


if __name__ == "__main__":
    main()
