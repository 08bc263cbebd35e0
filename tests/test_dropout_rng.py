"""Dropout decisions (reference preencoder.py:233 and :86-102: nn.Dropout(0.1) inside ConvBlock2D / ConvBlock): statistical
quality of the counter-based streams of csrc/common.h (drop_words: refiner / residual-block pointwise kernels) and
csrc/convblock2d.hip (DropStream: one bit-sliced stream per pixel).

The round-4 review found that the marginal rate was the only property under test and that two seeds which differ only in their
high word gave correlated masks (5.7e-3; 0.26 for the top bit - the seed's high word was XOR-ed into a GF(2)-linear generator
after the hash).  Bounds here, at 2^23 decisions (one sigma of an empirical correlation: 3.5e-4):

  * |rho| <= 2e-3 between the masks of consecutive ops.next_seed() values, of seeds that differ in one bit (low word, high word,
    top bit) and of seeds that differ only in the high word;
  * |rho| <= 2e-3 between neighbouring decisions inside one mask: lag 1, lag 2, the next group of 8 (pointwise); the next hidden
    unit, the next word of 32 units, the next pixel (ConvBlock2D);
  * the keep count per group of 8 has the binomial variance (no hidden structure inside a group).

CPU: the numpy model below (the definition, word for word).  GPU: the kernels' actual masks equal the model bit for bit
(pointwise) and meet the same bounds (both kernels)."""
import numpy as np
import pytest
import torch

U = np.uint32
BOUND = 2e-3
P_DROP = 0.1
THRESH16 = int(P_DROP * 65536.0 + 0.5)


def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> U(16)
    x = (x * U(0x7feb352d)).astype(np.uint32)
    x ^= x >> U(15)
    x = (x * U(0x846ca68b)).astype(np.uint32)
    x ^= x >> U(16)
    return x


def _seed_keys(seed):
    lo, hi = np.array([seed & 0xFFFFFFFF], dtype=np.uint32), np.array([seed >> 32], dtype=np.uint32)
    return (lo ^ _mix32(hi ^ U(0x85ebca6b)))[0], _mix32(hi + _mix32(lo ^ U(0xc2b2ae35)))[0]


def model_keep_mask(seed, n_groups):
    """csrc/common.h drop_words + Drop8::keep for groups 0 .. n_groups-1 (< 2^32): bool (n_groups, 8), True = kept."""
    with np.errstate(over="ignore"):
        k0, k1 = _seed_keys(seed)
        s = _mix32(_mix32(np.arange(n_groups, dtype=np.uint64).astype(np.uint32) ^ k0) + k1)
        s = np.where(s == 0, U(0x6d2b79f5), s)
        w = (s * U(0x9E3779B9)).astype(np.uint32)
        halves = []
        for _ in range(4):
            s ^= s << U(13)
            s ^= s >> U(17)
            s ^= s << U(5)
            w = (w + U(0x9E3779B9)).astype(np.uint32)
            word = (s + w).astype(np.uint32)
            halves += [word & U(0xFFFF), word >> U(16)]
    return np.stack(halves, axis=1) >= THRESH16


def corr(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(((a * b).mean() - a.mean() * b.mean()) / np.sqrt(a.var() * b.var()))


def _next_seeds(n):
    """What mqgan_amd.ops.manual_seed(123) + n calls of next_seed() produce (the package is not imported on CPU-only boxes for
    this: the two lines are its definition)."""
    base = (123 * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return [(base + i * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF for i in range(1, n + 1)]


SEED_PAIRS = [("one bit, low word", 1), ("one bit, bit 31", 1 << 31), ("one bit, high word", 1 << 32), ("top bit", 1 << 63),
              ("high word only", 5 << 32), ("high word only, all ones", 0xFFFFFFFF << 32)]


def _check_masks(get, what):
    """get(seed) -> bool array (n_groups, 8) of 2^23 decisions."""
    seeds = _next_seeds(4)
    masks = [get(s) for s in seeds]
    for m in masks:
        assert abs(m.mean() - (1.0 - THRESH16 / 65536.0)) < 1e-3, (what, m.mean())
    for i in range(3):
        r = corr(masks[i], masks[i + 1])
        assert abs(r) <= BOUND, f"{what}: consecutive next_seed() values {i + 1}, {i + 2} correlate at {r:.2e}"
    for name, d in SEED_PAIRS:
        r = corr(masks[2], get(seeds[2] ^ d))
        assert abs(r) <= BOUND, f"{what}: seeds that differ by {name} correlate at {r:.2e}"
    k = masks[0]
    flat = k.ravel()
    for name, r in (("lag 1", corr(flat[:-1], flat[1:])), ("lag 2", corr(flat[:-2], flat[2:])), ("next group", corr(k[:-1], k[1:])),
                    ("lag 4096 groups", corr(k[:-4096], k[4096:]))):
        assert abs(r) <= BOUND, f"{what}: neighbouring decisions ({name}) correlate at {r:.2e}"
    p = 1.0 - k.mean()
    v = k.sum(axis=1).var()
    assert abs(v - 8 * p * (1 - p)) <= 0.01, f"{what}: keep count per group of 8 has variance {v:.4f}, binomial {8 * p * (1 - p):.4f}"


def test_drop_words_model_cross_seed_and_neighbour_correlations():
    _check_masks(lambda s: model_keep_mask(s, 1 << 20), "model")


def test_round4_seeding_would_fail_these_bounds():
    """The bounds have teeth: the r4 seeding (seed's high word XOR-ed in after the hash) misses them by two orders of magnitude."""
    def r4_mask(seed, n_groups):
        with np.errstate(over="ignore"):
            g = np.arange(n_groups, dtype=np.uint64).astype(np.uint32)
            s = _mix32(g ^ U(seed & 0xFFFFFFFF)) ^ U(seed >> 32)
            s = np.where(s == 0, U(0x6d2b79f5), s)
            w = (s * U(0x9E3779B9)).astype(np.uint32)
            halves = []
            for _ in range(4):
                s ^= s << U(13)
                s ^= s >> U(17)
                s ^= s << U(5)
                w = (w + U(0x9E3779B9)).astype(np.uint32)
                word = (s + w).astype(np.uint32)
                halves += [word & U(0xFFFF), word >> U(16)]
        return np.stack(halves, axis=1) >= THRESH16
    s = _next_seeds(3)[2]
    assert abs(corr(r4_mask(s, 1 << 20), r4_mask(s ^ (1 << 63), 1 << 20))) > 0.1


# ------------------------------------------------------------------------------------------------------------ the kernels
@pytest.mark.gpu
def test_pointwise_dropout_masks_equal_the_model_and_meet_the_bounds():
    from gpu_util import DEV
    import __graft_entry__ as g
    g.build()
    from mqgan_amd import ops
    ops.manual_seed(123)
    assert [ops.next_seed() for _ in range(4)] == _next_seeds(4)
    x = torch.ones(1 << 11, 1 << 12, device=DEV)          # 2^23 elements, C = 4096: the vector path, 8 elements per group

    def kernel_mask(seed):
        y = ops.PointwiseFn.apply(x, None, None, None, None, None, (ops.ACT_NONE, False, False, (1, 1, 1), P_DROP, int(seed)))
        return (y != 0).view(-1, 8).cpu().numpy()

    s = _next_seeds(1)[0]
    np.testing.assert_array_equal(kernel_mask(s), model_keep_mask(s, 1 << 20))
    np.testing.assert_array_equal(kernel_mask(s ^ (1 << 63)), model_keep_mask(s ^ (1 << 63), 1 << 20))
    # bf16 storage: same groups, same decisions
    yh = ops.PointwiseFn.apply(x.to(torch.bfloat16), None, None, None, None, None, (ops.ACT_NONE, False, False, (1, 1, 1), P_DROP, int(s)))
    np.testing.assert_array_equal((yh != 0).view(-1, 8).cpu().numpy(), model_keep_mask(s, 1 << 20))
    _check_masks(kernel_mask, "pointwise kernel")


@pytest.mark.gpu
def test_convblock2d_dropout_bits_cross_seed_and_neighbour_correlations():
    """ConvBlock2D's decisions as the forward kernel stored them: [C0 / 32][pixels] words, bit b of word j = hidden unit 32 j + b of
    pixel (b, t, c) of the one-channel C0 x T image (every pixel has C0 hidden units)."""
    from gpu_util import DEV
    import __graft_entry__ as g
    g.build()
    from mqgan_amd import ops
    torch.manual_seed(0)
    B, T, C0 = 2, 1024, 64                                  # 131,072 pixels x 64 hidden units = 2^23 decisions
    npix = B * T * C0
    x = torch.randn(B, T, C0, device=DEV)
    mask = torch.zeros(B, T, dtype=torch.uint8, device=DEV)
    wdw, bdw = torch.randn(25, device=DEV) * 0.2, torch.zeros(1, device=DEV)
    w1, b1 = torch.randn(C0, device=DEV), torch.zeros(C0, device=DEV)
    w2, b2 = torch.randn(C0, device=DEV).requires_grad_(True), torch.zeros(1, device=DEV)

    def bits(seed):
        y = ops.ConvBlock2DFn.apply(x, mask, wdw, bdw, w1, b1, w2, b2, P_DROP, int(seed))
        words = y.grad_fn.saved_tensors[-1].view(torch.int32).view(C0 // 32, npix).cpu().numpy().astype(np.uint32)
        k = (words[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & U(1)      # [word][pixel][unit in word]
        return k.astype(bool)

    seeds = _next_seeds(4)
    ks = [bits(s) for s in seeds]
    for k in ks:
        assert abs(k.mean() - (1.0 - THRESH16 / 65536.0)) < 1e-3, k.mean()
    for i in range(3):
        r = corr(ks[i], ks[i + 1])
        assert abs(r) <= BOUND, f"ConvBlock2D: consecutive next_seed() values correlate at {r:.2e}"
    for name, d in SEED_PAIRS:
        r = corr(ks[2], bits(seeds[2] ^ d))
        assert abs(r) <= BOUND, f"ConvBlock2D: seeds that differ by {name} correlate at {r:.2e}"
    k = ks[0]
    for name, r in (("next unit", corr(k[:, :, :-1], k[:, :, 1:])), ("next word of 32 units", corr(k[:-1], k[1:])),
                    ("next pixel", corr(k[:, :-1], k[:, 1:])), ("next frame", corr(k[:, :-C0], k[:, C0:]))):
        assert abs(r) <= BOUND, f"ConvBlock2D: neighbouring decisions ({name}) correlate at {r:.2e}"
    per_pixel = k.sum(axis=(0, 2))                              # kept units per pixel: binomial(64, 0.9)
    p = 1.0 - k.mean()
    assert abs(per_pixel.var() - C0 * p * (1 - p)) <= 0.25, (per_pixel.var(), C0 * p * (1 - p))
