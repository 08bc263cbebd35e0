"""Worst-case rounding geometry for the screened codebook search (csrc/vq_screen.hip): the inputs its error bound exists for.

The screen orders codes by bf16 scores a(n,k) = -cn(k)/2 + <bf16(x_n), bf16(c_k)>; the exact winner k* is only guaranteed to
satisfy a(k*) >= max_k a(k) - E(n), E the bound on |score - dist| with score = -2a (header of vq_screen.hip).  Random data
sits 10-30x inside the bound (Cauchy-Schwarz), so a threshold that is too small by a factor of two passes every random test
(it did: ADVICE r5).  Here the bf16 rounding errors of x and of the two competing codes are made to line up:

  * the row's components are split into two groups; in group A every |x_i| is just BELOW a bf16 midpoint (RNE rounds the
    magnitude down by almost half an ulp), in group B just ABOVE one (rounds up);
  * code c1 lives on group A with components just below a midpoint, code c2 on group B with components just above:
    <xh, c1h> underestimates <x, c1> by almost (2^-7)|x_A||c1|, <xh, c2h> overestimates <x, c2> by as much;
  * c2 is then pulled away by a small component so that c1 wins the exact comparison by a hair.

The bf16 scores prefer c2 by up to ~0.69 E: outside E/2, inside E (for these rows the measured-residual bound of r6 and the
worst-case bound of r3-r5 agree to 3 %: every component's residual IS half an ulp).  `cases()` yields such (x, codebook) sets over several
dimensions, magnitudes and sign patterns, with random filler codes around them; `bf16_gap()` is the numpy model of the
screen's arithmetic the tests use to show where each case sits relative to the bound.
"""
import numpy as np


def bf16_rne(a: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def screen_bound(x: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """E(n) as l2_screen_prep_kernel forms it (r6): from the measured rounding residuals |x - bf16(x)| and max_k |c - bf16(c)|,
    the threshold on a = -score / 2 being the WHOLE bound on |score - dist|."""
    d = x.shape[1]
    f64 = np.float64
    nx = np.sqrt((x.astype(f64) ** 2).sum(1)) * 1.001
    ex = np.sqrt(((x - bf16_rne(x)).astype(f64) ** 2).sum(1)) * 1.001
    cm = np.sqrt((cb.astype(f64) ** 2).sum(1).max()) * 1.001
    ecm = np.sqrt(((cb - bf16_rne(cb)).astype(f64) ** 2).sum(1).max()) * 1.001
    return (2.02 * (ex * cm + 1.004 * nx * ecm) + 2.02 * (d * 2.0 ** -21) * nx * cm + (2.0 ** -20 + d * 2.0 ** -23) * cm * cm
            + 1e-30).astype(np.float32)


def worst_case_bound(x: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """The r3-r5 bound: bf16's worst-case relative error 2^-8 per operand instead of the measured residuals."""
    d = x.shape[1]
    nx = np.sqrt((x.astype(np.float64) ** 2).sum(1)) * 1.001
    cm = np.sqrt((cb.astype(np.float64) ** 2).sum(1).max()) * 1.001
    delta = 1.02 * 2.0 ** -7 + d * 2.0 ** -21
    return (2.02 * delta * nx * cm + (2.0 ** -20 + d * 2.0 ** -23) * cm * cm + 1e-30).astype(np.float32)


def bf16_gap(x: np.ndarray, cb: np.ndarray, winners: np.ndarray) -> np.ndarray:
    """max_k a(n,k) - a(n, winner_n) under the screen's arithmetic (bf16 operands, exact products, wide accumulation)."""
    xh, ch = bf16_rne(x).astype(np.float64), bf16_rne(cb).astype(np.float64)
    cn = (cb.astype(np.float64) ** 2).sum(1)
    a = xh @ ch.T - 0.5 * cn[None, :]
    return a.max(1) - a[np.arange(x.shape[0]), winners]


def _one(rng, d, scale_x, scale_c, pull):
    """One row and its two competing codes (see the module docstring).  Magnitudes are powers of two times (1 + 2^-8 -+ 2^-20):
    1 + 2^-8 is the midpoint between the bf16 neighbours 1 and 1 + 2^-7."""
    lo = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)           # rounds DOWN to 1
    hi = np.float32(1.0 + 2.0 ** -8 + 2.0 ** -20)           # rounds UP to 1 + 2^-7
    perm = rng.permutation(d)
    ga, gb = perm[: d // 2], perm[d // 2: 2 * (d // 2)]
    sg = rng.choice([-1.0, 1.0], size=d).astype(np.float32)
    x = np.zeros(d, np.float32)
    x[ga], x[gb] = lo, hi
    c1, c2 = np.zeros(d, np.float32), np.zeros(d, np.float32)
    c1[ga] = lo
    c2[gb] = hi
    c2[gb[0]] = np.float32(hi * np.float32(1.0 - pull))    # pull c2 away: c1 wins the exact comparison
    return (x * sg * np.float32(scale_x)), (c1 * sg * np.float32(scale_c)), (c2 * sg * np.float32(scale_c))


def cases(seed=0):
    """Yields (name, x (n, d) f32, codebook (k, d) f32).  Every row's two competing codes sit among random filler codes of
    smaller norm (they neither win nor raise max|c|)."""
    rng = np.random.default_rng(seed)
    for d in (8, 16, 64, 256):
        for sx, sc in ((1.0, 1.0), (2.0 ** -10, 2.0 ** -10), (2.0 ** 20, 2.0 ** -3)):
            rows, codes = [], []
            for pull in (3e-3, 1e-2, 3e-2):
                for _ in range(16):
                    x, c1, c2 = _one(rng, d, sx, sc, pull)
                    rows.append(x)
                    codes += [c1, c2]
            k_fill = 200
            fill = (rng.standard_normal((k_fill, d)) * (0.3 * sc)).astype(np.float32)
            cb = np.concatenate([np.stack(codes), fill], 0)
            cb = cb[rng.permutation(cb.shape[0])]
            yield f"d{d}_sx{sx:g}_sc{sc:g}", np.stack(rows), np.ascontiguousarray(cb)
