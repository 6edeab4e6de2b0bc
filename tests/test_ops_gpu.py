"""Operator-level parity: each HIP kernel, called through the C ABI, against the
CPU oracle (fp64 restatement of the reference op) on seeded inputs.

Tolerance (fp32 path, BASELINE.json: max |delta| < 1e-3 on the final image):
per-op budget  max|gpu - f64| <= 2e-5 * max(1, max|ref|)  -- two orders below
the end-to-end bar, a few times fp32 round-off of a K~3000 dot product.
"""
import math

import numpy as np
import pytest
import torch

from oracle import sd_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def _rng(seed):
    return np.random.default_rng(seed)


def _check(got, ref64, what, rtol=RTOL):
    ref = np.asarray(ref64, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    err = np.abs(got - ref)
    bound = rtol * max(1.0, float(np.abs(ref).max()))
    idx = np.unravel_index(int(err.argmax()), err.shape)
    assert err.max() <= bound, (f"{what}: max|d|={err.max():.3e} > {bound:.3e} at {idx} "
                                f"(got {got[idx]:.6f}, ref {ref[idx]:.6f}); mean|d|={err.mean():.3e}")


def _t(a):
    return torch.from_numpy(np.asarray(a)).double()


# ---- GroupNorm (+SiLU): groupnorm/mod.rs:53-82, silu.rs:14-16 -----------------------------
@pytest.mark.parametrize("n,c,h,w", [(2, 320, 16, 16), (1, 1920, 8, 8), (1, 128, 32, 32), (2, 2560, 8, 8),
                                     (1, 960, 16, 16), (1, 32, 4, 4), (2, 640, 1, 1), (1, 512, 64, 64)])
@pytest.mark.parametrize("silu", [False, True])
def test_group_norm(sd_ops, n, c, h, w, silu):
    g = _rng(c * 7 + h)
    x = (g.standard_normal((n, c, h, w)) * 1.7 + 0.9).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(c)).astype(np.float32)
    beta = (0.1 * g.standard_normal(c)).astype(np.float32)
    got = sd_ops.op_group_norm(x, gamma, beta, 32, 1e-5, silu)
    ref = O.group_norm(_t(x), _t(gamma), _t(beta), 32, 1e-5)
    if silu:
        ref = O.silu(ref)
    _check(got, ref.numpy(), f"group_norm{(n, c, h, w)} silu={silu}")


@pytest.mark.parametrize("mean,std,shape", [(30.0, 0.05, (1, 320, 32, 32)), (-250.0, 0.5, (2, 640, 16, 16)),
                                            (1000.0, 1.0, (1, 128, 64, 64)), (5.0, 1e-3, (1, 1920, 8, 8))])
def test_group_norm_large_mean(sd_ops, mean, std, shape):
    """|mean| >> std: the reference is two-pass (u = x - mean; mean(u^2), groupnorm/mod.rs:75-82), so it does not
    care; a sum / sum-of-squares kernel loses everything here.  The shifted statistics + float-float mean of
    k_norm.hip hold the per-op bound every other operator has (2e-5 * max|ref|)."""
    g = _rng(5)
    c = shape[1]
    x = (g.standard_normal(shape) * std + mean).astype(np.float32)
    # per-channel offsets inside a group as well (channels of one group with different means)
    x += (g.standard_normal((1, c, 1, 1)) * 3 * std).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(c)).astype(np.float32)
    beta = (0.1 * g.standard_normal(c)).astype(np.float32)
    got = sd_ops.op_group_norm(x, gamma, beta, 32, 1e-5, False)
    ref = O.group_norm(_t(x), _t(gamma), _t(beta), 32, 1e-5).numpy()
    _check(got, ref, f"group_norm large mean {mean} +- {std}")


# ---- LayerNorm: unet/mod.rs:523-525 --------------------------------------------------------
@pytest.mark.parametrize("rows,c", [(64, 160), (257, 320), (100, 640), (33, 1280), (5, 2048)])
def test_layer_norm(sd_ops, rows, c):
    g = _rng(rows + c)
    x = (g.standard_normal((rows, c)) * 2 - 0.5).astype(np.float32)
    gamma = (1 + 0.1 * g.standard_normal(c)).astype(np.float32)
    beta = (0.1 * g.standard_normal(c)).astype(np.float32)
    got = sd_ops.op_layer_norm(x, gamma, beta, 1e-5)
    ref = O.layer_norm(_t(x), _t(gamma), _t(beta), 1e-5)
    _check(got, ref.numpy(), f"layer_norm({rows},{c})")


# ---- Conv2d: all hot-path variants and every tile configuration -----------------------------
CONV_CASES = [
    # n, cin, h, w, cout, k, stride, ups
    (2, 320, 16, 16, 320, 3, 1, 0),
    (1, 4, 16, 16, 320, 3, 1, 0),     # conv_in: generic-K path (Cin = 4)
    (1, 320, 16, 16, 4, 3, 1, 0),     # conv_out: N = 4
    (1, 128, 16, 16, 3, 3, 1, 0),     # VAE conv_out: N = 3 (scalar epilogue)
    (1, 4, 8, 8, 4, 1, 1, 0),         # post_quant_conv 1x1, K = 4
    (2, 320, 16, 16, 320, 3, 2, 0),   # Downsample stride 2
    (1, 640, 8, 8, 640, 3, 1, 1),     # Upsample: nearest-2x folded into the gather
    (2, 640, 16, 16, 320, 1, 1, 0),   # 1x1 skip
    (1, 960, 8, 8, 640, 3, 1, 0),
    (1, 2560, 8, 8, 1280, 3, 1, 0),   # deepest level, K = 23040 (split-K)
    (1, 256, 24, 40, 128, 3, 1, 0),   # non-square, M not a tile multiple
    (3, 64, 5, 7, 96, 3, 1, 0),       # ragged everything
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(sd_ops, case):
    n, cin, h, w, cout, k, stride, ups = case
    g = _rng(hash(case) % (2 ** 31))
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    sd_ops.set_option("gemm_tile", "auto")
    sd_ops.set_option("splitk", 0)
    got = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
    xin = _t(x)
    if ups:
        xin = O.upsample2x(xin)
    ref = O.conv2d(xin, (_t(wt), _t(b)), stride=stride, padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv2d{case}")


@pytest.mark.parametrize("tile", list(range(10)) + [100, 101, 102, 103])   # 100+: the 8-wave LDS-DMA kernel's tiles
@pytest.mark.parametrize("splitk", [1, 3, 8])
def test_conv2d_all_tiles(sd_ops, tile, splitk):
    """Every tile configuration x split-K on one awkward shape (M, N not tile multiples)."""
    n, cin, h, w, cout = 2, 96, 13, 11, 208
    g = _rng(1000 + tile)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        sd_ops.set_option("gemm_tile", tile)
        sd_ops.set_option("splitk", splitk)
        got = sd_ops.op_conv2d(x, wt, b)
    finally:
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    ref = O.conv2d(_t(x), (_t(wt), _t(b)), padding=1)
    _check(got, ref.numpy(), f"conv tile={tile} splitk={splitk}")


def test_conv2d_splitk_repeatable(sd_ops):
    """The split-K reduce kernel sums the k slices in a fixed order (several lanes per output, fixed shuffle order): repeated launches of a
    24-slice GEMM are bit-identical."""
    n, cin, h, w, cout = 1, 1280, 8, 8, 320
    g = _rng(4321)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        sd_ops.set_option("gemm_tile", 0)
        sd_ops.set_option("splitk", 24)
        outs = [sd_ops.op_conv2d(x, wt, b) for _ in range(6)]
    finally:
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    _check(outs[0], O.conv2d(_t(x), (_t(wt), _t(b)), padding=1).numpy(), "conv split-K 24")


XCASES = [
    # (n, cin, h, w, cout, k, stride, ups): several M tiles with a ragged last one, N tails, every conv flavour
    (2, 128, 23, 19, 320, 3, 1, 0), (1, 64, 40, 36, 200, 3, 1, 0), (2, 192, 16, 16, 640, 1, 1, 0), (1, 128, 33, 31, 128, 3, 2, 0),
    (1, 64, 12, 20, 384, 3, 1, 1), (1, 96, 9, 7, 100, 3, 1, 0),
]


@pytest.mark.parametrize("tile", [100, 101, 102, 103])
@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("case", XCASES)
def test_conv2d_large_tiles(sd_ops, tile, splitk, case):
    """k_gemm2x.hip: 256-row, 8-wave fp32 tiles staged by LDS-DMA (tile 100 + x)."""
    n, cin, h, w, cout, k, stride, ups = case
    g = _rng(4000 + tile + 7 * splitk + cin + cout)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        sd_ops.set_option("gemm_tile", tile)
        sd_ops.set_option("splitk", splitk)
        got = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
    finally:
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    xin = O.upsample2x(_t(x)) if ups else _t(x)
    ref = O.conv2d(xin, (_t(wt), _t(b)), stride=stride, padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv large tile={tile} splitk={splitk} {case}")


STILES = [200, 201, 202, 203, 204, 205]


# gemm3x_variant bits -- 0: next k tile's DMA in one block behind the barrier; 1: scalar residual subtractions; 2: two LDS stages on the
# 128-row tiles too (default: three); 4: s_setprio 1 for waves 4-7
SPLIT_VARIANTS = [0, 1, 2, 6]


@pytest.mark.parametrize("variant", SPLIT_VARIANTS)
@pytest.mark.parametrize("tile", STILES)
@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("case", XCASES)
def test_conv2d_split_bf16_tiles(sd_ops, tile, splitk, case, variant):
    """k_gemm3x.hip: fp32 operands as exact sums of three bf16 terms, six partial products on the bf16 matrix pipe
    (tile 200 + x) -- held to the same bar as the fp32-MFMA kernels."""
    n, cin, h, w, cout, k, stride, ups = case
    g = _rng(5000 + tile + 7 * splitk + cin + cout)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        sd_ops.set_option("gemm3x_variant", variant)
        sd_ops.set_option("gemm_tile", tile)
        sd_ops.set_option("splitk", splitk)
        got = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
    finally:
        sd_ops.set_option("gemm3x_variant", 0)
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    xin = O.upsample2x(_t(x)) if ups else _t(x)
    ref = O.conv2d(xin, (_t(wt), _t(b)), stride=stride, padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv split-bf16 tile={tile} splitk={splitk} variant={variant} {case}")


SHORT_K_CASES = [
    # (n, cin, h, w, cout, k, splitk): one to four k tiles per slice -- the pipelined loop runs its prologue and the dead-stage
    # re-fetch of the last tile right next to each other here
    (1, 32, 16, 16, 64, 1, 1), (1, 64, 16, 16, 320, 1, 1), (1, 64, 16, 16, 320, 1, 2), (1, 96, 12, 12, 160, 1, 1), (1, 96, 12, 12, 160, 1, 3),
    (2, 32, 8, 8, 128, 3, 1), (2, 32, 8, 8, 128, 3, 3), (2, 32, 8, 8, 128, 3, 9), (1, 128, 20, 20, 100, 1, 1), (1, 128, 20, 20, 100, 1, 2),
]


@pytest.mark.parametrize("variant", [2])
@pytest.mark.parametrize("tile", STILES)
@pytest.mark.parametrize("case", SHORT_K_CASES)
def test_conv2d_split_bf16_short_k(sd_ops, tile, case, variant):
    """k_gemm3x.hip with 1 ... 4 k tiles per split-K slice, every tile shape."""
    n, cin, h, w, cout, k, splitk = case
    g = _rng(6000 + tile + 7 * splitk + cin + cout)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        sd_ops.set_option("gemm3x_variant", variant)
        sd_ops.set_option("gemm_tile", tile)
        sd_ops.set_option("splitk", splitk)
        got = sd_ops.op_conv2d(x, wt, b)
    finally:
        sd_ops.set_option("gemm3x_variant", 0)
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    ref = O.conv2d(_t(x), (_t(wt), _t(b)), padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv split-bf16 short K tile={tile} variant={variant} {case}")


def test_conv2d_split_bf16_placement_variants_bit_identical(sd_ops):
    """The placement switches of the plain k loop (variant bits 2 / 4: how many LDS stages on the 128-row tiles, wave priority)
    change WHEN operands arrive, not the arithmetic: every tile shape gives bit-identical results under all of them."""
    n, cin, h, w, cout = 2, 320, 24, 24, 320
    g = _rng(8118)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    try:
        for tile in STILES:
            for splitk in (1, 4):
                sd_ops.set_option("gemm_tile", tile)
                sd_ops.set_option("splitk", splitk)
                outs = {}
                for variant in (2, 6, 18):
                    sd_ops.set_option("gemm3x_variant", variant)
                    outs[variant] = sd_ops.op_conv2d(x, wt, b)
                for variant, o in outs.items():
                    assert np.array_equal(o, outs[2]), f"tile {tile} splitk {splitk}: variant {variant} differs from variant 2"
    finally:
        sd_ops.set_option("gemm3x_variant", 0)
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)


def test_conv2d_split_bf16_is_fp32_accurate(sd_ops):
    """What the three-way split costs, measured: on a K = 11520 convolution whose inputs span ten binary orders of
    magnitude per channel, the split kernel's error against the fp64 oracle is of the size of the fp32-MFMA kernel's own
    (both are fp32 accumulations of essentially exact products; measured 2.8e-6 against 4.6e-6 of max|ref|), inside the
    2e-5 bar -- no bf16-sized (4e-3) or two-term-sized (1.5e-5 per product) error appears."""
    n, cin, h, w, cout = 1, 1280, 16, 16, 320
    g = _rng(777)
    x = (g.standard_normal((n, cin, h, w)) * np.exp2(g.integers(-5, 6, (1, cin, 1, 1)))).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9) * np.exp2(g.integers(-3, 4, (cout, 1, 1, 1)))).astype(np.float32)
    ref = O.conv2d(_t(x), (_t(wt), None), padding=1).numpy()
    scale = np.abs(ref).max()
    errs = {}
    try:
        for name, tile in (("fp32 mfma 128x320x", 103), ("split 128x320s", 201), ("split 256x128s", 202)):
            sd_ops.set_option("gemm_tile", tile)
            sd_ops.set_option("splitk", 1)
            got = sd_ops.op_conv2d(x, wt, None)
            errs[name] = float(np.abs(got - ref).max() / scale)
    finally:
        sd_ops.set_option("gemm_tile", "auto")
        sd_ops.set_option("splitk", 0)
    print("max |gpu - fp64| / max|ref|, K = 11520: " + ", ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    assert errs["split 128x320s"] < 3.0 * errs["fp32 mfma 128x320x"] + 1e-7
    assert errs["split 256x128s"] < 3.0 * errs["fp32 mfma 128x320x"] + 1e-7
    assert max(errs.values()) < 1e-5


def test_conv2d_split_bf16_exact_on_small_integers(sd_ops):
    """Integers up to 2^16 need the low planes (bf16 alone keeps 8 bits): products and sums below 2^24 are exact in the split
    kernel, bit for bit."""
    g = _rng(31337)
    x = g.integers(-300, 301, (1, 64, 9, 9)).astype(np.float32)
    wt = g.integers(-40, 41, (64, 64, 3, 3)).astype(np.float32)
    try:
        sd_ops.set_option("gemm_tile", 205)
        got = sd_ops.op_conv2d(x, wt, None)
    finally:
        sd_ops.set_option("gemm_tile", "auto")
    ref = O.conv2d(_t(x), (_t(wt), None), padding=1).numpy()
    assert np.abs(ref).max() < 2 ** 24
    assert np.array_equal(got.astype(np.float64), ref)


def test_conv2d_asymmetric_weights_not_transposed(sd_ops):
    """A = identity-like check with an asymmetric kernel: catches tap (ky,kx) swaps."""
    x = np.zeros((1, 32, 6, 6), np.float32)
    x[0, 0, 2, 3] = 1.0
    wt = np.zeros((32, 32, 3, 3), np.float32)
    wt[5, 0] = np.arange(9, dtype=np.float32).reshape(3, 3) + 1
    got = sd_ops.op_conv2d(x, wt, None)
    ref = O.conv2d(_t(x), (_t(wt), None), padding=1).numpy()
    assert np.array_equal(got, ref.astype(np.float32))


# ---- Linear ----------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cin,cout", [(154, 768, 320), (20, 320, 1280), (512, 320, 2560), (1, 1280, 1280), (77, 64, 160)])
def test_linear(sd_ops, rows, cin, cout):
    g = _rng(rows * 3 + cout)
    x = g.standard_normal((rows, cin)).astype(np.float32)
    wt = (g.standard_normal((cin, cout)) / math.sqrt(cin)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    got = sd_ops.op_linear(x, wt, b)
    _check(got, O.linear(_t(x), _t(wt), _t(b)).numpy(), f"linear({rows},{cin},{cout})")
    got = sd_ops.op_linear(x, wt, None)
    _check(got, O.linear(_t(x), _t(wt), None).numpy(), f"linear nobias({rows},{cin},{cout})")


# ---- GEGLU, timestep embedding -------------------------------------------------------------------
def test_geglu(sd_ops):
    g = _rng(3)
    proj = (g.standard_normal((37, 2 * 640)) * 2).astype(np.float32)
    got = sd_ops.op_geglu(proj)
    p = _t(proj)
    ref = p[:, :640] * O.gelu_erf(p[:, 640:])
    _check(got, ref.numpy(), "geglu", rtol=1e-5)


@pytest.mark.parametrize("t", [999, 949, 49, 0, 1])
def test_timestep_embedding(sd_ops, t):
    got = sd_ops.op_timestep_embedding(t, 320)
    ref32 = O.timestep_embedding(t, 320, 10000, torch.float32).numpy()
    # cos/sin of arguments up to 999 rad in f32: the argument itself carries ~6e-5 abs error,
    # so compare with the f32 oracle (same f32 argument), not the f64 one
    # freq = exp(j*coef) may differ by 1 ulp between f32 exp implementations; the argument t*freq
    # (up to 999 rad) amplifies that to t * 2^-23
    assert np.abs(got - ref32).max() < 2e-6 + 1.3e-7 * t


# ---- qkv_attention: attention.rs:5-45 ----------------------------------------------------------------
ATTN_CASES = [
    # n, nq, nk, n_state, n_head
    (2, 256, 256, 320, 8),    # d = 40 self
    (1, 1024, 1024, 320, 8),
    (2, 64, 64, 640, 8),      # d = 80
    (1, 256, 256, 1280, 8),   # d = 160
    (2, 256, 77, 320, 8),     # cross, T = 77 (ragged last tile)
    (2, 64, 2, 1280, 8),      # cross, Tu = 2 (unconditional context)
    (1, 100, 37, 160, 4),     # ragged queries and keys
    (1, 4, 4, 640, 4),        # 2x2 level of the tiny model
    (3, 2048, 512, 320, 8),   # >= 384 8-wave workgroups -> the 8-wave instance (64x64 UNet level regime)
    (6, 2100, 300, 160, 4),   # 8-wave, ragged queries and keys
    (3, 2048, 77, 640, 8),    # 8-wave, d = 80, cross-attention length
    (1, 64, 64, 128, 1),      # unfused path (VAE-style single wide head)
    (1, 256, 256, 512, 1),
]


@pytest.mark.parametrize("case", ATTN_CASES)
def test_qkv_attention(sd_ops, case):
    n, nq, nk, c, heads = case
    g = _rng(hash(case) % (2 ** 31))
    q = g.standard_normal((n, nq, c)).astype(np.float32)
    k = g.standard_normal((n, nk, c)).astype(np.float32)
    v = g.standard_normal((n, nk, c)).astype(np.float32)
    got = sd_ops.qkv_attention(q, k, v, None, heads)
    ref = O.qkv_attention(_t(q), _t(k), _t(v), None, heads)
    _check(got, ref.numpy(), f"qkv_attention{case}")


@pytest.mark.parametrize("case", [(2, 1024, 1024, 320, 8), (1, 300, 77, 640, 8), (1, 200, 333, 160, 4)])
def test_qkv_attention_split_is_fp32_accurate(sd_ops, case):
    """k_attn_split.hip (fp32 q/k/v as three bf16 terms each, six partial products on the bf16 matrix pipe, fp32 softmax)
    against k_attn.hip (fp32 matrix instruction): same bar, and an error of the same size against the fp64 oracle."""
    n, nq, nk, c, heads = case
    g = _rng(4321 + nq)
    q = (g.standard_normal((n, nq, c)) * 1.5).astype(np.float32)
    k = (g.standard_normal((n, nk, c)) * 1.5).astype(np.float32)
    v = (g.standard_normal((n, nk, c)) * np.exp2(g.integers(-4, 5, (1, 1, c)))).astype(np.float32)
    ref = O.qkv_attention(_t(q), _t(k), _t(v), None, heads).numpy()
    errs = {}
    try:
        for mode in (1, 0):
            sd_ops.set_option("attn_split", mode)
            got = sd_ops.qkv_attention(q, k, v, None, heads)
            _check(got, ref, f"qkv_attention{case} attn_split={mode}")
            errs[mode] = float(np.abs(got - ref).max() / np.abs(ref).max())
    finally:
        sd_ops.set_option("attn_split", 1)
    print(f"attention {case}: max |gpu - fp64| / max|ref|: split {errs[1]:.2e}, fp32 mfma {errs[0]:.2e}")
    assert errs[1] < 3.0 * errs[0] + 2e-7


@pytest.mark.parametrize("case", [(2, 4096, 4096, 320, 8), (2, 1024, 77, 320, 8), (1, 300, 77, 320, 8), (1, 517, 1000, 320, 8), (1, 64, 2, 40, 1), (2, 1024, 1024, 320, 8)])
@pytest.mark.parametrize("splits", [0, 3])
def test_qkv_attention_packed_tail_d40(sd_ops, case, splits):
    """k_attn_split.hip at d = 40 (round 5): columns 32..39 of q / k ride a PACKED k step ([K_h | K_m], [K_h | K_l] against the query columns in both lane halves) and
    columns 32..39 of v a packed output tile (V_h, V_m, V_l as row groups, summed per lane after the key loop).  Against the six-instruction form (attn_pack_tail=0) and
    the fp64 oracle: the same bar, an error no larger than the old form's (it adds three partial products), and every column -- the packed ones separately -- within it;
    v's columns carry different magnitudes so that a row group landing in the wrong column would show."""
    n, nq, nk, c, heads = case
    g = _rng(5000 + nq + nk + splits)
    q = (g.standard_normal((n, nq, c)) * 1.5).astype(np.float32)
    k = (g.standard_normal((n, nk, c)) * 1.5).astype(np.float32)
    v = (g.standard_normal((n, nk, c)) * np.exp2(g.integers(-4, 5, (1, 1, c))) + np.arange(c, dtype=np.float32) % 7).astype(np.float32)
    ref = O.qkv_attention(_t(q), _t(k), _t(v), None, heads).numpy()
    errs = {}
    try:
        sd_ops.set_option("attn_kv_splits", splits)
        for mode in (3, 2, 1, 0):
            sd_ops.set_option("attn_pack_tail", mode)
            got = sd_ops.qkv_attention(q, k, v, None, heads)
            _check(got, ref, f"qkv_attention{case} attn_pack_tail={mode} attn_kv_splits={splits}")
            d = c // heads
            tail = np.abs(got - ref).reshape(n, nq, heads, d)[..., 32:].max()
            errs[mode] = (float(np.abs(got - ref).max() / np.abs(ref).max()), float(tail / np.abs(ref).max()))
    finally:
        sd_ops.set_option("attn_pack_tail", "default")
        sd_ops.set_option("attn_kv_splits", 0)
    print(f"attention {case} S={splits}: max |gpu - fp64| / max|ref| (all columns, columns 32..39): packed + log2 softmax {errs[3]}, log2 softmax {errs[2]}, packed {errs[1]}, round-4 form {errs[0]}")
    for mode in (1, 2, 3):
        assert errs[mode][0] < 1.5 * errs[0][0] + 1e-7 and errs[mode][1] < 1.5 * errs[0][1] + 1e-7, mode


def test_qkv_attention_causal_mask(sd_ops):
    """attn_decoder_mask (attention.rs:47-56) as the additive mask."""
    n, s, c, heads = 1, 77, 320, 8
    g = _rng(77)
    q, k, v = (g.standard_normal((n, s, c)).astype(np.float32) for _ in range(3))
    mask = np.triu(np.full((s, s), -np.inf, np.float32), 1)
    got = sd_ops.qkv_attention(q, k, v, mask, heads)
    ref = O.qkv_attention(_t(q), _t(k), _t(v), _t(mask), heads)
    _check(got, ref.numpy(), "qkv_attention causal")


def test_qkv_attention_spiked_scores(sd_ops):
    """Online-softmax rescale: a key in a LATER tile dominates (forces the max to jump)."""
    n, nq, nk, c, heads = 1, 64, 256, 320, 8
    g = _rng(9)
    q = g.standard_normal((n, nq, c)).astype(np.float32)
    k = g.standard_normal((n, nk, c)).astype(np.float32)
    v = g.standard_normal((n, nk, c)).astype(np.float32)
    k[0, 200] = q[0, 7] * 6.0   # huge score for query 7 at key 200 (4th tile)
    k[0, 3] = q[0, 9] * 6.0     # and in the first tile for query 9
    got = sd_ops.qkv_attention(q, k, v, None, heads)
    ref = O.qkv_attention(_t(q), _t(k), _t(v), None, heads)
    _check(got, ref.numpy(), "qkv_attention spiked")


def test_bad_arguments_fail_loudly(sd_ops):
    from stable_diffusion_burn_amd import SdmiError
    x = np.zeros((1, 48, 4, 4), np.float32)   # Cin = 48: neither < 32 nor a multiple of 32
    w = np.zeros((32, 48, 3, 3), np.float32)
    with pytest.raises(SdmiError):
        sd_ops.op_conv2d(x, w, None)
    with pytest.raises(SdmiError):
        sd_ops.set_option("no_such_option", 1)


# ---- GEGLU::forward, fused into the projection GEMM's epilogue --------------------------------------------------
@pytest.mark.parametrize("rows,cin,hidden", [(700, 320, 1280), (300, 64, 200), (2048, 320, 1280), (513, 128, 384)])
@pytest.mark.parametrize("fuse", [2, 3, 4, 5, 6, 0])   # 2 / 3: fused, 256x128 / 256x256 tiles; 4 / 5 / 6: the split kernel's 128x256s / 256x128s / 128x128s; 0: GEMM + gate kernel
def test_geglu_forward(sd_ops, rows, cin, hidden, fuse):
    """GEGLU::forward (unet/mod.rs:579-591).  fuse = 2: the gate runs in the large-tile GEMM's epilogue (value and gate
    fragments interleaved per wave, no [rows, 2 hidden] tensor); 0: projection GEMM + gate kernel."""
    g = _rng(rows + hidden + fuse)
    x = g.standard_normal((rows, cin)).astype(np.float32)
    w = (g.standard_normal((cin, 2 * hidden)) / math.sqrt(cin)).astype(np.float32)
    b = g.standard_normal(2 * hidden).astype(np.float32)
    try:
        sd_ops.set_option("geglu_fuse", fuse)
        got = sd_ops.op_geglu_forward(x, w, b, hidden)
    finally:
        sd_ops.set_option("geglu_fuse", 1)
    proj = _t(x) @ _t(w) + _t(b)
    ref = (proj[:, :hidden] * O.gelu_erf(proj[:, hidden:])).numpy()
    _check(got, ref, f"geglu_forward ({rows},{cin},{hidden}) fuse={fuse}")


# ---- round 5: the gate in the PLANE GEMM's epilogue, value / gate rows split by wave column (tiles with an odd fragment count per wave qualify) -----------
@pytest.mark.parametrize("rows,cin,hidden", [(700, 320, 1280), (300, 64, 224), (2048, 320, 1280), (513, 128, 384), (8192, 320, 1280)])
@pytest.mark.parametrize("tile", ["auto", 300, 301, 302, 303, 304, 305, 306, 307])
def test_geglu_forward_plane_tiles_wave_column_pairs(sd_ops, rows, cin, hidden, tile):
    """GEGLU::forward (unet/mod.rs:579-591) as the fp32 model runs it since round 5: x arrives as bf16 planes, the projection runs on a k_gemm3p.hip tile whose
    wave columns [0, WN / 2) multiply by the value rows and [WN / 2, WN) by the gate rows of the SAME outputs, the two waves of a pair exchange their fragments
    through LDS in the epilogue (k_gemm_epi.hpp, geglu = 2), and the gated result leaves as planes (geglu_fuse = 7, joined back exactly) and as fp32 (8) --
    the [rows, 2 hidden] tensor and the gate kernel's launch are gone.  Same operator bar as every fp32 GEMM."""
    g = _rng(rows + hidden + (0 if tile == "auto" else tile))
    x = g.standard_normal((rows, cin)).astype(np.float32)
    w = (g.standard_normal((cin, 2 * hidden)) / math.sqrt(cin)).astype(np.float32)
    b = g.standard_normal(2 * hidden).astype(np.float32)
    proj = _t(x) @ _t(w) + _t(b)
    ref = (proj[:, :hidden] * O.gelu_erf(proj[:, hidden:])).numpy()
    try:
        sd_ops.set_option("gemm_tile", tile)
        outs = []
        for fuse in (7, 8):
            sd_ops.set_option("geglu_fuse", fuse)
            got = sd_ops.op_geglu_forward(x, w, b, hidden)
            _check(got, ref, f"geglu_forward plane tile {tile} ({rows},{cin},{hidden}) fuse={fuse}")
            outs.append(got)
        np.testing.assert_array_equal(outs[0], outs[1])     # the planes ARE the fp32 result, split
    finally:
        sd_ops.set_option("geglu_fuse", 1)
        sd_ops.set_option("gemm_tile", "auto")


# ---- round 5: key slices of the fp32 attention kernels + the merge launch (option attn_kv_splits; automatic where the query-tile grid leaves CUs idle) ----------
KV_SPLIT_CASES = [
    # (n, nq, nk, c, heads): d = 80 on k_attn_split.hip (the 32 x 32 level at batch 1), d = 160 on k_attn.hip (16 x 16), d = 40, ragged key counts, fewer tiles than slices
    (2, 1024, 1024, 640, 8), (2, 256, 256, 1280, 8), (1, 300, 333, 640, 8), (1, 200, 700, 160, 4), (2, 64, 64, 1280, 8), (1, 128, 77, 320, 8), (1, 70, 100, 640, 4),
]


@pytest.mark.parametrize("case", KV_SPLIT_CASES)
@pytest.mark.parametrize("splits", [0, 2, 3, 4, 8])
def test_qkv_attention_key_slices(sd_ops, case, splits):
    """qkv_attention (attention.rs:5-45) with the keys cut into S slices that run as S x the workgroups and are merged by a second launch:
    out = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s -- the online-softmax identity across workgroups instead of across a workgroup's tiles.  Same bar as the
    unsliced kernels; a spiked key in a late slice exercises the merge's rescale; splits = 0 is the engine's own rule; with the plane output (gemm_planes, the model's form)
    the result must EQUAL the fp32 output's planes."""
    n, nq, nk, c, heads = case
    g = _rng(7000 + nq + nk + c + splits)
    q = g.standard_normal((n, nq, c)).astype(np.float32)
    k = g.standard_normal((n, nk, c)).astype(np.float32)
    v = (g.standard_normal((n, nk, c)) * np.exp2(g.integers(-3, 4, (1, 1, c)))).astype(np.float32)
    k[0, nk - 5] = q[0, 7] * 5.0        # a dominating key in the LAST slice for query 7, and one in the first slice for query 9
    k[0, 2] = q[0, 9] * 5.0
    ref = O.qkv_attention(_t(q), _t(k), _t(v), None, heads).numpy()
    try:
        sd_ops.set_option("attn_kv_splits", splits)
        got = sd_ops.qkv_attention(q, k, v, None, heads)
        _check(got, ref, f"qkv_attention{case} attn_kv_splits={splits}")
        sd_ops.set_option("gemm_planes", 0)
        plain = sd_ops.qkv_attention(q, k, v, None, heads)     # fp32 rows instead of joined planes: the same values
        np.testing.assert_array_equal(got, plain)
    finally:
        sd_ops.set_option("attn_kv_splits", 0)
        sd_ops.set_option("gemm_planes", "default")
