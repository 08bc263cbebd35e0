#!/usr/bin/env python3
"""`trace_around.py kernel_trace.csv SUBSTRING [before after]`: from a rocprofv3 --kernel-trace CSV, print every launch whose
kernel name contains SUBSTRING together with the launches around it (name, grid, duration) - which transform / GEMM launches of a
layer are slow in situ."""
import csv, sys
path, sub = sys.argv[1], sys.argv[2]
nb, na = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 3)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if sub in r["Kernel_Name"]]
shown = set()
for i in hits:
    for j in range(max(0, i - nb), min(len(rows), i + na + 1)):
        if j in shown:
            continue
        shown.add(j)
        r = rows[j]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3 if j else 0.0
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        print(f"{'>>' if j == i else '  '} {j:6d} {dur:9.1f} us  gap {gap:7.1f}  grid {grid:>10s} wg {r.get('Workgroup_Size_X', '?'):>4s}  {r['Kernel_Name'][:120]}")
    print()
