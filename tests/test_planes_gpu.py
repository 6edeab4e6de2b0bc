"""k_gemm3p.hip -- the fp32 GEMM / convolution on the bf16 matrix pipe with BOTH operands as three bf16 planes (tile 300 + x) -- against the
fp64 oracle, at the bar every fp32 operator holds (tests/test_ops_gpu.py: 2e-5 max(1, |ref|)).  Reference arithmetic: Burn's Conv2d / Linear
at unet/mod.rs:397,425,468,479,553,580,645-651,716,726,729 and autoencoder/mod.rs:516-523,568-604.  Also here: the behaviour of the
producer-side split (k_split3.hpp, s3_split1) at the edges of the fp32 range, which include/sdmi.h documents."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd_oracle as O  # noqa: E402

PTILES = [300, 301, 302, 303, 304, 305, 306, 307, 308]
XCASES = [
    # (n, cin, h, w, cout, k, stride, ups): several M tiles with a ragged last one, N tails, every conv flavour
    (2, 128, 23, 19, 320, 3, 1, 0), (1, 64, 40, 36, 200, 3, 1, 0), (2, 192, 16, 16, 640, 1, 1, 0), (1, 128, 33, 31, 128, 3, 2, 0),
    (1, 64, 12, 20, 384, 3, 1, 1), (1, 96, 9, 7, 100, 3, 1, 0),
]
SHORT_K_CASES = [
    # (n, cin, h, w, cout, k, splitk): one to four k tiles per slice (prologue and dead-stage re-fetch next to each other)
    (1, 32, 16, 16, 64, 1, 1), (1, 64, 16, 16, 320, 1, 1), (1, 64, 16, 16, 320, 1, 2), (1, 96, 12, 12, 160, 1, 1), (1, 96, 12, 12, 160, 1, 3),
    (2, 32, 8, 8, 128, 3, 1), (2, 32, 8, 8, 128, 3, 3), (2, 32, 8, 8, 128, 3, 9), (1, 128, 20, 20, 100, 1, 1), (1, 128, 20, 20, 100, 1, 2),
]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def _check(got, ref, what, rel=2e-5):
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    err = np.abs(got.astype(np.float64) - ref).max()
    tol = rel * max(1.0, np.abs(ref).max())
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e}"


class _Forced:
    """gemm_planes = 2 (plane tiles where asked for) + a forced tile / split-K, restored on exit"""

    def __init__(self, sd, tile, splitk=0):
        self.sd, self.tile, self.splitk = sd, tile, splitk

    def __enter__(self):
        self.sd.set_option("gemm_planes", 2)
        self.sd.set_option("gemm_tile", self.tile)
        self.sd.set_option("splitk", self.splitk)
        return self.sd

    def __exit__(self, *a):
        self.sd.set_option("gemm_tile", "auto")
        self.sd.set_option("splitk", 0)
        self.sd.set_option("gemm_planes", "default")


@pytest.mark.parametrize("tile", PTILES)
@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("case", XCASES)
def test_conv2d_plane_tiles(sd_ops, tile, splitk, case):
    n, cin, h, w, cout, k, stride, ups = case
    g = np.random.default_rng(9000 + tile + 7 * splitk + cin + cout)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    with _Forced(sd_ops, tile, splitk):
        got = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
        again = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
    xin = O.upsample2x(_t(x)) if ups else _t(x)
    ref = O.conv2d(xin, (_t(wt), _t(b)), stride=stride, padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv planes tile={tile} splitk={splitk} {case}")
    assert np.array_equal(got, again)


@pytest.mark.parametrize("tile", PTILES)
@pytest.mark.parametrize("case", SHORT_K_CASES)
def test_conv2d_plane_tiles_short_k(sd_ops, tile, case):
    n, cin, h, w, cout, k, splitk = case
    g = np.random.default_rng(9500 + tile + 7 * splitk + cin + cout)
    x = g.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    with _Forced(sd_ops, tile, splitk):
        got = sd_ops.op_conv2d(x, wt, b)
    ref = O.conv2d(_t(x), (_t(wt), _t(b)), padding=1 if k == 3 else 0)
    _check(got, ref.numpy(), f"conv planes short K tile={tile} {case}")


@pytest.mark.parametrize("tile", PTILES)
@pytest.mark.parametrize("splitk", [1, 3])
def test_lean_fp32_epilogue_equals_the_general_one_bit_for_bit(sd_ops, tile, splitk):
    """Round 6 (k_gemm_epi.hpp): interior single-sample tiles take an epilogue with every address computed once per tile; gemm3x_variant bit 6 sends them through the general form.
    Same values in the same order: the two must agree bit for bit -- on shapes with interior AND edge tiles, one sample per tile and several (8 x 8 images: the lean form must
    step aside), 1x1 and 3x3, with split-K slabs."""
    for case in [(2, 64, 32, 32, 320, 1), (2, 64, 32, 32, 320, 3), (5, 96, 8, 8, 160, 3), (1, 128, 40, 36, 200, 1), (3, 32, 16, 16, 128, 3)]:
        n, cin, h, w, cout, k = case
        g = np.random.default_rng(9700 + tile + 7 * splitk + cin + cout)
        x = g.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k)).astype(np.float32)
        b = g.standard_normal(cout).astype(np.float32)
        outs = {}
        try:
            with _Forced(sd_ops, tile, splitk):
                for variant in (2, 66):
                    sd_ops.set_option("gemm3x_variant", variant)
                    outs[variant] = sd_ops.op_conv2d(x, wt, b)
        finally:
            sd_ops.set_option("gemm3x_variant", "default")
        assert np.array_equal(outs[2], outs[66]), f"tile {tile} splitk {splitk} {case}: lean and general epilogue differ"
        ref = O.conv2d(_t(x), (_t(wt), _t(b)), padding=1 if k == 3 else 0)
        _check(outs[2], ref.numpy(), f"conv planes lean epilogue tile={tile} splitk={splitk} {case}")


def test_plane_kernel_is_fp32_accurate_and_exact_on_small_integers(sd_ops):
    """The K = 11520, ten-binades-per-channel convolution of test_conv2d_split_bf16_is_fp32_accurate: the plane kernel's error against fp64 is
    of the size of the fp32 matrix instruction's own; integers whose products and sums stay below 2^24 come out bit-exact (they need the
    low planes: bf16 alone keeps 8 bits)."""
    n, cin, h, w, cout = 1, 1280, 16, 16, 320
    g = np.random.default_rng(777)
    x = (g.standard_normal((n, cin, h, w)) * np.exp2(g.integers(-5, 6, (1, cin, 1, 1)))).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9) * np.exp2(g.integers(-3, 4, (cout, 1, 1, 1)))).astype(np.float32)
    ref = O.conv2d(_t(x), (_t(wt), None), padding=1).numpy()
    scale = np.abs(ref).max()
    xi = g.integers(-300, 301, (1, 64, 9, 9)).astype(np.float32)
    wi = g.integers(-40, 41, (64, 64, 3, 3)).astype(np.float32)
    refi = O.conv2d(_t(xi), (_t(wi), None), padding=1).numpy()
    assert np.abs(refi).max() < 2 ** 24
    with _Forced(sd_ops, 103, 1):
        e_mfma = float(np.abs(sd_ops.op_conv2d(x, wt, None) - ref).max() / scale)
    errs = {}
    for tile in PTILES:
        with _Forced(sd_ops, tile, 1):
            errs[tile] = float(np.abs(sd_ops.op_conv2d(x, wt, None) - ref).max() / scale)
            assert np.array_equal(sd_ops.op_conv2d(xi, wi, None).astype(np.float64), refi), f"tile {tile}: small integers not exact"
    print(f"max |gpu - fp64| / max|ref|, K = 11520: fp32 mfma {e_mfma:.2e}, planes " + ", ".join(f"{t}: {e:.2e}" for t, e in errs.items()))
    for tile, e in errs.items():
        assert e < 1e-5 and e < 3.0 * e_mfma + 1e-7, f"tile {tile}: {e:.2e} (fp32 matrix instruction: {e_mfma:.2e})"


def test_plane_path_at_the_edges_of_the_fp32_range(sd_ops):
    """NaN, +-inf, values that round-to-nearest would carry to inf in bf16 (3.4e38, -FLT_MAX) and tiny values through
    the plane path (producer-side split s3_split1 + k_gemm3p.hip), next to the fp32 matrix instruction (tile 103).  What include/sdmi.h
    ("fp32 semantics of precision 0") promises, and this test pins:
      * outputs are finite exactly where the fp32 matrix instruction's are; NaN stays NaN;
      * an INFINITE operand gives a non-finite result, but possibly NaN where fp32 gives +-inf: the low-order partial products contain
        0 * inf.  (The split itself keeps inf: h = inf, m = l = 0 -- not the inf - inf = NaN of a naive residual.)
      * finite values up to FLT_MAX are exact: where round-to-nearest of h would overflow to inf, h is truncated instead;
      * below 2^-109 the low planes are flushed: correct to an absolute 2^-118 max|w| per term."""
    cin, cout, hw = 64, 64, 8
    x = np.zeros((1, cin, hw, hw), np.float32)
    wt = np.zeros((cout, cin, 1, 1), np.float32)
    wt[np.arange(cout), np.arange(cout) % cin, 0, 0] = 1.0                 # output channel c = input channel c
    wt[1, 1] = 2.0 ** -100
    wt[7, 7] = 3.0
    x[0, 0, 0, 0] = np.inf
    x[0, 0, 0, 1] = -np.inf
    x[0, 2, 1, 1] = np.nan
    x[0, 1, 2, 2] = np.float32(3.4e38)                                    # bf16 round-to-nearest of it is inf
    x[0, 1, 2, 3] = np.float32(-3.4028235e38)                             # -FLT_MAX
    x[0, 3, 3, 3] = np.float32(1e-38)
    x[0, 4, 3, 4] = np.float32(1e-44)                                     # fp32 subnormal
    x[0, 7, 4, 4] = np.float32(1.0000001)
    with _Forced(sd_ops, 103, 1):
        mfma = sd_ops.op_conv2d(x, wt, None)
    assert mfma[0, 0, 0, 0] == np.inf and mfma[0, 0, 0, 1] == -np.inf and np.isnan(mfma[0, 2, 1, 1])
    for tile in (300, 304):
        with _Forced(sd_ops, tile, 1):
            got = sd_ops.op_conv2d(x, wt, None)
        fin = np.isfinite(mfma)
        assert np.array_equal(np.isfinite(got), fin), "finite / non-finite pattern differs from the fp32 matrix instruction"
        assert np.isnan(got[np.isnan(mfma)]).all(), "NaN must stay NaN"
        assert not np.isfinite(got[0, 0, 0, 0]) and not np.isfinite(got[0, 0, 0, 1])
        assert got[0, 1, 2, 2] == np.float32(3.4e38) * np.float32(2.0 ** -100)
        assert got[0, 1, 2, 3] == np.float32(-3.4028235e38) * np.float32(2.0 ** -100)
        assert got[0, 7, 4, 4] == np.float32(1.0000001) * np.float32(3.0)
        assert abs(float(got[0, 3, 3, 3]) - 1e-38) <= 2.0 ** -118
        assert abs(float(got[0, 4, 3, 4]) - 1e-44) <= 2.0 ** -118
        assert np.abs(got[fin].astype(np.float64) - mfma[fin].astype(np.float64)).max() <= 2.0 ** -118


@pytest.mark.parametrize("rows,cin,cout", [(154, 768, 320), (20, 320, 1280), (512, 320, 2560), (1, 1280, 1280), (77, 64, 160)])
def test_linear_on_plane_tiles(sd_ops, rows, cin, cout):
    g = np.random.default_rng(9900 + rows + cin)
    x = g.standard_normal((rows, cin)).astype(np.float32)
    wt = (g.standard_normal((cin, cout)) / math.sqrt(cin)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    ref = (_t(x) @ _t(wt) + _t(b)).numpy()
    for tile in (303, 304):
        with _Forced(sd_ops, tile, 0):
            got = sd_ops.op_linear(x, wt, b)
        _check(got, ref, f"linear planes tile={tile} rows={rows} cin={cin} cout={cout}")


# ---- model level: every producer writing planes (option gemm_planes = 1) -----------------------------------------------------------
def _tiny_inputs(d, n, T):
    from stable_diffusion_burn_amd import synthetic as syn
    lat = np.stack([syn.initial_latent(i, d.latent_h, d.latent_w) for i in range(n)])
    ctx = np.stack([syn.cond_context(i, T, d.ctx_dim) for i in range(n)])
    return lat, ctx


class _Planes:
    def __init__(self, sd, on):
        self.sd, self.on = sd, on

    def __enter__(self):
        self.sd.set_option("gemm_planes", self.on)
        return self.sd

    def __exit__(self, *a):
        self.sd.set_option("gemm_planes", "default")


@pytest.mark.parametrize("t", [999, 49])
def test_unet_forward_with_producer_written_planes(sd_tiny, synth, tiny_dims, t):
    """UNet::forward (unet/mod.rs:109-143) with GroupNorm / LayerNorm / GEGLU / attention / GEMM epilogues writing bf16 planes and every
    eligible GEMM on k_gemm3p.hip: the model-level bar of tests/test_model_gpu.py, and agreement with the fp32-staged path to fp32 noise."""
    from stable_diffusion_burn_amd import synthetic as syn
    d = tiny_dims
    lat, ctx = _tiny_inputs(d, 2, 7)
    a = syn.alphas_cumprod()
    o32 = O.StableDiffusionOracle(synth, a, d, torch.float32)
    o64 = O.StableDiffusionOracle(synth, a, d, torch.float64)
    with _Planes(sd_tiny, 0):
        base = sd_tiny.unet.forward(lat, [t], ctx)
    with _Planes(sd_tiny, 1):
        got = sd_tiny.unet.forward(lat, [t], ctx)
        again = sd_tiny.unet.forward(lat, [t], ctx)
    r32 = o32.unet.forward(torch.from_numpy(lat), t, torch.from_numpy(ctx)).numpy()
    r64 = o64.unet.forward(torch.from_numpy(lat), t, torch.from_numpy(ctx)).numpy()
    e64 = np.abs(got.astype(np.float64) - r64).max()
    e32 = np.abs(r32.astype(np.float64) - r64).max()
    print(f"unet t={t} planes: |gpu-f64|={e64:.2e} |f32-f64|={e32:.2e} |planes - staged|={np.abs(got - base).max():.2e}")
    assert np.isfinite(got).all() and e64 <= max(1e-4, 2 * e32)
    assert np.array_equal(got, again)
    assert np.abs(got - base).max() <= 2e-5 * max(1.0, np.abs(r64).max())


def test_sample_image_with_producer_written_planes(sd_tiny, synth, tiny_dims):
    """sample_image (stablediffusion/mod.rs:51-160): 3 DDIM steps + VAE decode with planes on -- latent within 1e-3 of the fp32 oracle, u8 image within 1 LSB"""
    from stable_diffusion_burn_amd import synthetic as syn
    d = tiny_dims
    lat, ctx = _tiny_inputs(d, 1, 7)
    unc = syn.uncond_context(2, d.ctx_dim)
    ora = O.StableDiffusionOracle(synth, syn.alphas_cumprod(), d, torch.float32)
    ref_lat = ora.sample_latent(torch.from_numpy(ctx), torch.from_numpy(unc), 7.5, 3, torch.from_numpy(lat)).numpy()
    ref_img, _ = ora.latent_to_image(torch.from_numpy(ref_lat))
    with _Planes(sd_tiny, 1):
        got_lat = sd_tiny.sample_latent(ctx, unc, 7.5, 3, init_latent=lat)
        got_img = sd_tiny.sample_image(ctx, unc, 7.5, 3, init_latent=lat)
    dl = float(np.abs(got_lat - ref_lat).max())
    di = int(np.abs(got_img.astype(np.int16) - np.asarray(ref_img).astype(np.int16)).max())
    print(f"planes: max|latent - oracle| = {dl:.2e}; u8 max diff = {di}")
    assert np.isfinite(got_lat).all() and dl < 1e-3
    assert di <= 1


# ---- operator level: the plane-writing producers are the fp32 producers, bit for bit -------------------------------------------------
def test_plane_producers_equal_their_fp32_forms_bit_for_bit(sd_ops):
    """GroupNorm(+SiLU) (groupnorm/mod.rs:53-82, silu.rs:14-16), LayerNorm (unet/mod.rs:523-525), the GEGLU gate (unet/mod.rs:579-591) and
    qkv_attention (attention.rs:5-45) writing three bf16 planes instead of fp32: with option gemm_planes the operator-level entry points run
    the plane-writing kernels and join the planes back (h + m + l, exact), so the results must EQUAL the fp32-writing kernels' -- a wrong
    plane position, a dropped low plane or a mis-split value shows up as a difference."""
    g = np.random.default_rng(4242)
    x = (g.standard_normal((2, 320, 16, 16)) * np.exp2(g.integers(-6, 7, (1, 320, 1, 1)))).astype(np.float32)
    gam, bet = g.standard_normal(320).astype(np.float32), g.standard_normal(320).astype(np.float32)
    rows = (g.standard_normal((257, 640)) * 3 + 1).astype(np.float32)
    lg, lb = g.standard_normal(640).astype(np.float32), g.standard_normal(640).astype(np.float32)
    proj = g.standard_normal((100, 2 * 1280)).astype(np.float32)
    q, k, v = (g.standard_normal((2, 300, 320)).astype(np.float32) for _ in range(3))
    q160, k160, v160 = (g.standard_normal((1, 70, 640)).astype(np.float32) for _ in range(3))

    def run():
        return [sd_ops.op_group_norm(x, gam, bet, silu=True), sd_ops.op_group_norm(x, gam, bet, silu=False), sd_ops.op_layer_norm(rows, lg, lb),
                sd_ops.op_geglu(proj), sd_ops.qkv_attention(q, k, v, None, 8), sd_ops.qkv_attention(q160, k160, v160, None, 4)]
    with _Planes(sd_ops, 0):
        base = run()
    with _Planes(sd_ops, 1):
        got = run()
    for name, a, b in zip(["gn+silu", "gn", "layer_norm", "geglu", "attention d=40", "attention d=160"], base, got):
        assert np.isfinite(b).all(), name
        if name == "geglu":   # (the two gate kernels are compiled separately: hipcc contracts a * gelu(g) differently, one fp32 rounding apart)
            assert np.abs(a - b).max() <= 2.0 ** -22 * np.abs(a).max(), f"geglu: {np.abs(a - b).max():.3e}"
            continue
        assert np.array_equal(a, b), f"{name}: plane-writing kernel differs from the fp32-writing one by {np.abs(a - b).max():.3e}"
