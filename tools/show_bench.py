#!/usr/bin/env python3
"""Compact view of a bench.py JSON line.  usage: tools/show_bench.py file.json [n_kernels]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def show(d, title):
    n = nk
    r = d["roofline"]
    st = r.get("step", {})
    print(f"{title}: {d['value']:.0f} {d['unit']}  {d['ms_per_step']:.1f} ms/step  finite={d['losses_finite']}  dominant {r['kernel']} "
          f"{r['achieved']:.1f} {r['unit']} frac {r['frac']:.3f}  conv share {r['conv_share_of_step_time']:.3f}  "
          f"step {st.get('TFLOP/s', 0):.1f} TF/s frac {st.get('frac', 0):.3f}  traffic {r.get('traffic')}")
    tot = 0.0
    for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        tot += v["ms_per_step"]
        if n > 0:
            print(f"    {k:44s} {v['ms_per_step']:8.2f} ms {v['TFLOPs']:7.1f} TF  x{v['launches_per_step']:.0f}")
            n -= 1
    print(f"    conv total {tot:.1f} ms, rest {d['ms_per_step'] - tot:.1f} ms")


show(d, "main")
for k, v in d.get("configs", {}).items():
    show(v, k)
for k, v in d.get("other_engines", {}).items():
    print(f"other engine {k}: {v['ms_per_step']:.1f} ms/step")
for k, v in d.get("vq", {}).items():
    print(f"vq {k}: {v['ms']:.4f} ms frac {v['frac']:.3f}")
if "cpu_baseline" in d:
    print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"][:200])
