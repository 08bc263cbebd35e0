#!/usr/bin/env python3
"""Per-kernel register / LDS / spill summary of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py mqgan_amd/csrc/igemm.hip [name-filter]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" [")[0]] = v
for name, r in rows.items():
    if filt in name:
        print(f"{name[:70]:70s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} spill={r.get('VGPRs Spill')} occ={r.get('Occupancy')} lds={r.get('LDS Size')}")
