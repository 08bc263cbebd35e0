#!/usr/bin/env python3
"""The L2 codebook search alone - the f32-MFMA kernel and the bf16-screened search - for tools/pmc_h2.sh counters:
python tools/vq_micro.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for k, d in ((8192, 512), (2048, 512), (1024, 256)):
    x = torch.randn(rows, d, device="cuda")
    cb = torch.randn(k, d, device="cuda")
    cn = ops.l2_codebook_norms(cb)
    ws = ops._ws_bytes(ops.L().mq_l2_argmin_workspace(rows), x)
    for _ in range(3):
        ops.l2_argmin(x, cb, cn, workspace=ws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.l2_argmin(x, cb, cn, workspace=ws)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"l2_argmin rows {rows} codes {k} d {d}: {ms:.4f} ms  {2.0 * rows * k * d / ms / 1e9:.1f} TF/s", flush=True)
    prep = ops.l2_screen_prepare(cb)
    ws2 = ops._ws_bytes(ops.L().mq_l2_screen_workspace(rows, k, d), x)
    for _ in range(3):
        ops.l2_argmin_screened(x, cb, prep, workspace=ws2)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        ops.l2_argmin_screened(x, cb, prep, workspace=ws2)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    emitted, cap, over = ops.l2_screen_stats(ws2, rows, k, d)
    print(f"l2_argmin_screened rows {rows} codes {k} d {d}: {ms:.4f} ms  {2.0 * rows * k * d / ms / 1e9:.1f} TF/s (algorithmic); "
          f"{emitted / rows:.2f} list entries per row of {cap // rows}, {over} rows searched exhaustively", flush=True)
