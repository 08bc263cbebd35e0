"""The C restatement (oracle/vq_ref.c) against fixture G1 and against itself."""
import numpy as np
import torch

from conftest import load_golden
from oracle import nn as O
from oracle import vq


def test_fsq_c_matches_reference_g1():
    g = load_golden("g1_fsq.npz")
    levels = g["levels"].tolist()
    codes, idx = vq.fsq_forward(g["z"], levels)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(codes, g["codes"])
    np.testing.assert_array_equal(vq.fsq_decode(np.arange(1000), levels), g["decode_all"])
    g2 = load_golden("g1b_fsq_idx16k.npz")
    _, idx2 = vq.fsq_forward(g2["z"], levels)
    np.testing.assert_array_equal(idx2, g2["idx"].astype(np.int32))


def test_fsq_is_l2_argmin_over_implicit_codebook():
    """SURVEY section 8a row Q3 pin (i): on the FSQ grid the generic L2 nearest
    neighbour search reproduces the FSQ index (away from rounding boundaries)."""
    g = load_golden("g1_fsq.npz")
    x = g["bounded_norm"].reshape(-1, 4)
    idx = vq.l2_argmin(x, g["implicit_codebook"], threads=4)
    ref = g["idx"].reshape(-1)
    safe = g["margin"].reshape(-1, 4).min(axis=1) > 1e-4
    np.testing.assert_array_equal(idx[safe], ref[safe])
    assert safe.mean() > 0.99


def test_l2_argmin_is_a_true_minimiser():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((300, 64)).astype(np.float32)
    cb = rng.standard_normal((257, 64)).astype(np.float32)
    idx, dist = vq.l2_argmin(x, cb, threads=2, return_dist=True)
    d2 = ((x[:, None, :].astype(np.float64) - cb[None].astype(np.float64)) ** 2).sum(-1)
    best = d2.min(axis=1)
    chosen = d2[np.arange(300), idx]
    assert np.all(chosen <= best * (1 + 1e-5) + 1e-5)
    # ties -> lowest index
    cb2 = np.concatenate([cb[:5], cb[:5]], 0)
    idx2 = vq.l2_argmin(cb[:5], cb2)
    np.testing.assert_array_equal(idx2, np.arange(5))
    # thread split does not change the answer
    np.testing.assert_array_equal(vq.l2_argmin(x, cb, threads=1), idx)


def test_fsq_thresholds_exact_vs_host_torch_tanh():
    """What 'bit-exact FSQ indices' is measured against.  The product's default thresholds come from tanh correctly rounded
    (the definition vq_ref.c restates); the reference runs torch.tanh on f32 (quantizer.py:113).  On this host torch's f32
    tanh is monotone within 64 ulps of every threshold (checked by make_fsq_desc(..., 'torch')) and the two threshold sets
    agree except for a few entries ONE ulp apart - so the C oracle and torch CPU can differ only when z is exactly one of
    those floats.  For levels [8,5,5,5] and z ~ N(0, 2^2) that is about 3e-8 per row."""
    import math
    from mqgan_amd.quantizer import _f2o, make_fsq_desc
    for levels in ([8, 5, 5, 5], [8, 8, 5, 5, 5], [7, 3], [2, 9, 4]):
        a, b = make_fsq_desc(levels), make_fsq_desc(levels, "torch")
        p_row = 0.0
        for j, L in enumerate(levels):
            for k in range(L - 1):
                ta, tb = np.float32(a.thresh[j][k]), np.float32(b.thresh[j][k])
                gap = abs(_f2o(ta) - _f2o(tb))
                assert gap <= 1, (levels, j, k, ta, tb)
                if gap:
                    ulp = float(np.spacing(np.abs(ta)))
                    p_row += ulp * math.exp(-float(ta) ** 2 / 8.0) / (2.0 * math.sqrt(2 * math.pi))
                    # the float between the thresholds is where the C oracle and torch CPU disagree
                    z = np.zeros((16, len(levels)), np.float32)
                    z[:, j] = min(ta, tb)
                    _, ic = vq.fsq_forward(z, levels)
                    _, it = O.fsq_forward(torch.from_numpy(z), levels)
                    assert (ic != it.numpy()).all()
        assert p_row < 1e-7, (levels, p_row)
        if levels == [8, 5, 5, 5]:
            assert 1e-8 < p_row < 6e-8                       # ~3e-8: 1-2 flips in 2^26 rows, as measured


def test_adversarial_rounding_cases_sit_between_half_and_the_whole_bound():
    """The generator of tests/vq_adversarial.py (inputs of the GPU test of the screened search): in every case some rows'
    exact winner trails the best bf16 score by MORE than half the bound - the threshold r5 shipped, which dropped them
    silently - and none by more than the bound itself (numpy model of the screen's arithmetic and of its E(n))."""
    import vq_adversarial as A
    n_half = 0
    for name, x, cb in A.cases():
        want = vq.l2_argmin(x, cb)
        gap = A.bf16_gap(x, cb, want)
        r = gap / A.screen_bound(x, cb)
        assert r.max() <= 0.75, (name, r.max())          # inside the bound, with the margin its slack factors give
        r5 = gap / (0.5 * A.worst_case_bound(x, cb))     # what r5 compared against
        assert (r5 > 1.0).sum() >= 5, (name, r5.max())   # ... and outside r5's halved threshold: it lost these rows
        n_half += int((r5 > 1.0).sum())
        assert np.all(A.screen_bound(x, cb) <= 1.01 * A.worst_case_bound(x, cb))
    assert n_half >= 200
    # random data: the measured-residual bound is about half the worst-case one
    rng = np.random.default_rng(5)
    x, cb = rng.standard_normal((64, 256)).astype(np.float32), rng.standard_normal((512, 256)).astype(np.float32)
    ratio = A.screen_bound(x, cb) / A.worst_case_bound(x, cb)
    assert 0.3 < ratio.min() and ratio.max() < 0.6, (ratio.min(), ratio.max())
