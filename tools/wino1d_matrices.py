#!/usr/bin/env python3
"""Transform matrices of one-dimensional Winograd F(m, 3) (Toom-Cook through the transposition principle) as exact
rationals, printed as the constexpr tables of csrc/wino1d.hip, with the f32 error of each point set against float64.

    y = A^T [ (G g) . (B^T d) ]      d: m + 2 input rows, g: 3 taps, y: m output rows

`python tools/wino1d_matrices.py` prints the tables for the point sets the kernels use; `--survey` measures alternatives."""
import sys
from fractions import Fraction as Fr

import numpy as np

POINTS = {2: [0, 1, -1], 4: [0, 1, -1, 2, Fr(-1, 2)], 5: [0, 1, -1, 2, -2, Fr(1, 2)]}


def matrices(m, pts, r=3):
    n = m + r - 1
    P = [Fr(p) for p in pts]
    assert len(P) == n - 1

    def vand(cols):
        rows = [[p ** j for j in range(cols)] for p in P]
        rows.append([Fr(0)] * (cols - 1) + [Fr(1)])           # the point at infinity: leading coefficient
        return rows
    V, Vm, Vr = vand(n), vand(m), vand(r)
    A = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if A[i][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        pv = A[c][c]
        A[c] = [x / pv for x in A[c]]
        for i in range(n):
            if i != c and A[i][c] != 0:
                f = A[i][c]
                A[i] = [x - f * y for x, y in zip(A[i], A[c])]
    Vinv = [row[n:] for row in A]
    AT = [[Vm[i][j] for i in range(n)] for j in range(m)]
    G = [row[:] for row in Vr]
    BT = [[Vinv[j][i] for j in range(n)] for i in range(n)]
    return AT, G, BT


def as_np(Mx):
    return np.array([[float(x) for x in row] for row in Mx])


def error(m, pts, C=256, K=64, taps=3, samples=2048, seed=0):
    AT, G, BT = (as_np(x) for x in matrices(m, pts))
    n = m + 2
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((samples, n, taps * C)).astype(np.float32)
    w = (rng.standard_normal((K, 3, taps * C)) * 0.05).astype(np.float32)
    win = np.stack([x[:, i:i + 3] for i in range(m)], 1)
    ref = np.einsum('sijc,kjc->sik', win.astype(np.float64), w.astype(np.float64))
    f = np.float32
    Vt = np.einsum('pn,snc->spc', BT.astype(f), x)
    U = np.einsum('pr,krc->kpc', G.astype(f), w)
    Mt = np.einsum('spc,kpc->spk', Vt, U)
    y = np.einsum('mp,spk->smk', AT.astype(f), Mt)
    direct = np.einsum('sijc,kjc->sik', win, w)
    sc = np.sqrt((ref ** 2).mean())
    e, ed = (y - ref) / sc, (direct - ref) / sc
    return float(np.sqrt((e ** 2).mean())), float(np.abs(e).max()), float(np.sqrt((ed ** 2).mean())), float(np.abs(ed).max())


def table(name, Mx):
    rows = ["{" + ", ".join(f"{float(x)!r}f" if x.denominator not in (1,) else f"{int(x)}.f" for x in row) + "}" for row in Mx]
    return f"    static constexpr float {name}[{len(Mx)}][{len(Mx[0])}] = {{" + ", ".join(rows) + "};"


if __name__ == "__main__":
    if "--survey" in sys.argv:
        for m, pts in [(2, [0, 1, -1]), (3, [0, 1, -1, 2]), (4, [0, 1, -1, 2, -2]), (4, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
                       (4, [0, 1, -1, 2, Fr(-1, 2)]), (5, [0, 1, -1, 2, -2, Fr(1, 2)]), (5, [0, 1, -1, 2, Fr(-1, 2), Fr(1, 2)]),
                       (6, [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2)])]:
            r = error(m, pts)
            print(f"F({m},3) points {[str(Fr(p)) for p in pts]}: rms {r[0]:.2e} max {r[1]:.2e}   (direct f32: rms {r[2]:.2e} max {r[3]:.2e})")
        sys.exit(0)
    for m, pts in POINTS.items():
        AT, G, BT = matrices(m, pts)
        print(f"// F({m}, 3), points {', '.join(str(Fr(p)) for p in pts)}, inf")
        for nm, Mx in (("BT", BT), ("G", G), ("AT", AT)):
            print(table(nm, Mx))
            print("    //", [[str(x) for x in row] for row in Mx])
