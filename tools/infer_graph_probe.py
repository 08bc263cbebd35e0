"""Can PreEncoder.encode / .decode be captured into a hipGraph (torch.cuda.graph over the ctypes launches), and what does a
replay cost against the eager call?  hifispeech architecture, random weights, eval mode, frozen weights."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import yaml  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd.scripted_preencoder import build_from_config, freeze_weights  # noqa: E402
from mqgan_amd.preencoder import sequence_mask  # noqa: E402

torch.cuda.set_device(0)
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "configs", "model_config_hifispeech.yaml")))
torch.manual_seed(0)
m = build_from_config(cfg).to("cuda").eval()
freeze_weights(m)
mel = cfg["model"]["mel_channels"]


def lat(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, T in ((1, 256), (1, 512), (4, 512), (32, 512)):
    x = torch.randn(B, T, mel, device="cuda") * 2 - 4
    lens = torch.full((B,), T, dtype=torch.long)
    lens[-1] = T - 7
    mask = sequence_mask(T, lens.cuda()).unsqueeze(1)
    with torch.no_grad():
        idx = m.encode(x, mask)
        y = m.decode(idx, mask)
        e_enc = lat(lambda: m.encode(x, mask))
        e_dec = lat(lambda: m.decode(idx, mask))
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    m.encode(x, mask); m.decode(idx, mask)
            torch.cuda.current_stream().wait_stream(s)
            ge_, gd_ = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ge_):
                idx_g = m.encode(x, mask)
            with torch.cuda.graph(gd_):
                y_g = m.decode(idx, mask)
            ge_.replay(); gd_.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(idx_g, idx)) and bool(torch.equal(y_g, y))
            g_enc, g_dec = lat(ge_.replay), lat(gd_.replay)
            print(f"B {B} T {T}: encode eager {e_enc:.2f} ms, graph {g_enc:.2f} ms; decode eager {e_dec:.2f} ms, graph {g_dec:.2f} ms; "
                  f"identical {same}", flush=True)
        except Exception as e:      # noqa: BLE001
            print(f"B {B} T {T}: encode eager {e_enc:.2f} ms, decode eager {e_dec:.2f} ms; capture failed: {type(e).__name__}: {e}", flush=True)
            break
