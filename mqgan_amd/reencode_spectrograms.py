"""Batch re-encoding of .npy spectrogram trees (reference reencode_spectrograms.py:8-108 and
reencode_spectrograms_from_checkpoint.py:9-108, same CLIs): load -> zero-pad to the batch maximum ->
encode -> decode -> trim -> save to the mirrored path; a failing batch is reported and skipped.

Batching.  The reference cuts the ``os.walk`` order into batches, so every batch is padded to its
longest item and - because padded frames leak into valid ones in the reference's model (ConvBlock2D writes its output
bias on padded frames, CBAM's max-pool and spatial gate are unmasked) - a file's result depends on
which neighbours it happened to be batched with.  ``bucket`` selects the policy here:
  * ``"sorted"``: files are sorted by length and cut into batches of neighbours whose padding waste stays
    under ``max_pad`` (fraction of the batch's frames).  Least padding work; results depend on padding exactly as the
    reference's do.
  * ``"exact"``: only files of identical length share a batch - no padding at all, so every file gets bit-for-bit the
    result of being processed alone, whatever else is in the tree.
  * ``"order"`` (default, as the reference-compatible entry points must behave): the reference's loop - directory order
    cut into batches, so the same tree gives the same batches, hence the same outputs, as the reference's script.
``sorted`` and ``exact`` are opt-in (``--bucket``): faster / padding-independent, but NOT the reference's batching.
Outputs are always written to the mirrored path of their input, whatever the processing order."""
import argparse
import os
import time

import numpy as np
import torch
import yaml

from .preencoder import get_pre_encoder, sequence_mask
from .scripted_preencoder import ScriptedPreEncoder, freeze_weights


def _find_npy(input_dir):
    return [os.path.join(root, f) for root, _, files in os.walk(input_dir) for f in files if f.endswith(".npy")]


def _length_of(path):
    """Frame count from the .npy header (no data read); None if the file is unreadable or not 2-D."""
    try:
        a = np.load(path, mmap_mode="r")
        return int(a.shape[0]) if a.ndim == 2 else None
    except Exception:
        return None


def plan_batches(files, lengths, batch_size, bucket="order", max_pad=0.1):
    """List of batches (lists of paths).  Files whose length is unknown go one per batch at the end (their failure must
    not take readable files down with it)."""
    if bucket == "order":
        return [files[i:i + batch_size] for i in range(0, len(files), batch_size)]
    known = sorted((L, i) for i, L in enumerate(lengths) if L is not None)      # by length, ties in directory order
    batches, cur, cur_sum = [], [], 0
    for L, i in known:
        if cur:
            longest = L                                                        # sorted: the newcomer is the longest
            waste = 1.0 - (cur_sum + L) / float(longest * (len(cur) + 1))
            split = len(cur) >= batch_size or (bucket == "exact" and L != lengths[cur[0]]) or (bucket == "sorted" and waste > max_pad)
            if split:
                batches.append([files[j] for j in cur])
                cur, cur_sum = [], 0
        cur.append(i)
        cur_sum += L
    if cur:
        batches.append([files[j] for j in cur])
    batches += [[files[i]] for i, L in enumerate(lengths) if L is None]
    return batches


def _run(encode_decode, input_dir, output_dir, batch_size, bucket="order", max_pad=0.1, stats=None):
    files = _find_npy(input_dir)
    if not files:
        print("Warning: No .npy files were found.")
        return 0
    if bucket not in ("sorted", "exact", "order"):
        raise ValueError(f"bucket must be 'sorted', 'exact' or 'order', got {bucket!r}")
    lengths = [None] * len(files) if bucket == "order" else [_length_of(p) for p in files]
    done, frames, padded, t0 = 0, 0, 0, time.time()
    for paths in plan_batches(files, lengths, batch_size, bucket, max_pad):
        try:
            specs = [np.load(p) for p in paths]
            lens = [s.shape[0] for s in specs]
            max_len = max(lens)
            batch = np.stack([np.concatenate([s, np.zeros((max_len - s.shape[0], s.shape[1]), dtype=s.dtype)], axis=0)
                              for s in specs])
            out = encode_decode(torch.from_numpy(batch).float(), lens)
            out = out.cpu().numpy()
            for j, p in enumerate(paths):
                rel = os.path.relpath(p, input_dir)
                dst = os.path.join(output_dir, rel)
                os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
                np.save(dst, out[j, :lens[j], :])
                done += 1
            frames += sum(lens)
            padded += max_len * len(lens)
        except Exception as e:           # reference: print and continue with the next batch (:83-85)
            print(f"\nCould not process batch starting with {paths[0]}. Error: {e}")
            continue
    if stats is not None:
        el = time.time() - t0
        stats.update(files=done, frames=frames, padded_frames=padded, seconds=el, frames_per_s=frames / el if el > 0 else 0.0,
                     pad_fraction=(1.0 - frames / padded) if padded else 0.0)
    return done


def reencode_spectrograms(model_path, input_dir, output_dir, device, batch_size, bucket="order", max_pad=0.1, stats=None):
    """ScriptedPreEncoder variant (model directory)."""
    try:
        model = ScriptedPreEncoder(model_path, device=device)
    except (FileNotFoundError, RuntimeError) as e:
        print(f"Error: Could not load the model. {e}")
        return 0
    return _run(lambda x, lens: model.decode(model.encode(x, lengths=lens), lengths=lens), input_dir, output_dir, batch_size, bucket,
                max_pad, stats)


def reencode_spectrograms_from_checkpoint(checkpoint_path, config, input_dir, output_dir, device, batch_size, bucket="order",
                                          max_pad=0.1, stats=None):
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
    return _run(enc_dec, input_dir, output_dir, batch_size, bucket, max_pad, stats)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Re-encode spectrograms with a PreEncoder on MI355X.")
    ap.add_argument("--model", type=str, default=None, help="model directory written by mqgan_amd.export")
    ap.add_argument("--checkpoint", type=str, default=None, help="raw training checkpoint (.pth) - needs --config")
    ap.add_argument("--config", type=str, default=None)
    ap.add_argument("--input_dir", type=str, required=True)
    ap.add_argument("--output_dir", type=str, required=True)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--bucket", type=str, default="order", choices=["order", "sorted", "exact"],
                    help="batching policy: the reference's directory order (default), length-sorted neighbours, or identical lengths only")
    ap.add_argument("--max_pad", type=float, default=0.1, help="'sorted': largest fraction of padded frames a batch may carry")
    a = ap.parse_args(argv)
    stats = {}
    if a.model:
        n = reencode_spectrograms(a.model, a.input_dir, a.output_dir, a.device, a.batch_size, a.bucket, a.max_pad, stats)
    elif a.checkpoint and a.config:
        with open(a.config) as f:
            n = reencode_spectrograms_from_checkpoint(a.checkpoint, yaml.safe_load(f), a.input_dir, a.output_dir, a.device,
                                                      a.batch_size, a.bucket, a.max_pad, stats)
    else:
        raise SystemExit("give --model DIR or --checkpoint FILE --config YAML")
    print(f"\nProcessing complete: {n} file(s) written to {a.output_dir}")
    if stats:
        print(f"{stats['frames']} frames in {stats['seconds']:.2f} s = {stats['frames_per_s']:.0f} mel-frames/s "
              f"({100 * stats['pad_fraction']:.1f} % of the processed frames were padding)")


if __name__ == "__main__":
    main()
