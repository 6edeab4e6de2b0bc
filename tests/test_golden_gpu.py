"""FULL-SIZE parity (SD v1.4 dims, BASELINE.json configs[0] and configs[1]) of the HIP path
against the committed golden fixtures produced by the CPU oracle
(tests/golden/gen_golden.py; fp32 and fp64 runs of oracle/sd_oracle.py on the
seeded synthetic weights / inputs of BASELINE.md section 3).

Stated tolerances (BASELINE.json north_star):
  * latents and decoded float RGB:  max |gpu - oracle_f32| < 1e-3  (absolute; final latent
    absmax is ~62 with untrained weights, so this is ~1.6e-5 relative), and the fp64 oracle as
    tie-breaker:  |gpu - f64| <= max(1e-3, 2 * |f32 - f64|)
  * u8 image: <= 1 LSB (the reference truncates, stablediffusion/mod.rs:96)
"""
from pathlib import Path

import numpy as np
import pytest

from stable_diffusion_burn_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"

# bf16 bars = 1.5 x the relative RMS measured on MI355X against the fp64 fixtures ("bar set after first
# measurement"); measured: UNet forward 1.0e-2, 20-step latent 0.9e-2, 50-step latent see test output, RGB 0.9e-2
BF16_BAR_UNET = 1.5e-2
BF16_BAR_LATENT20 = 1.4e-2
BF16_BAR_LATENT50 = 3.0e-2
BF16_BAR_RGB = 1.4e-2


# Round 6 -- the CHAINED error: the decode of the GPU's OWN reduced-precision latent against the fp64 image of the fp64 latent (latent error -> RGB error end to end), beside
# the decode-of-the-oracle-latent checks above.  Bars = 1.5 x first measurement on MI355X (printed by the tests; profiles/README.md r06c): the decoder is not a contraction --
# a latent off by 0.6 ... 5 % moves the image by more than the decoder's own bf16 error.
#   measured (profiles/r06c_pytest_golden_chained_checks.txt): bf16 S = 50 rel-RMS RGB 1.02-1.03e-2, u8 mean |d| 0.62-0.63 LSB; bf16 S = 20 1.21-1.22e-2, 0.67 LSB;
#   precision 2 default 5.4-5.5e-2, 2.05-2.10 LSB; fp8_linear = 1 8.0-8.2e-2, 2.98-3.05 LSB
BF16_BAR_RGB_CHAINED = {20: 1.85e-2, 50: 1.55e-2}   # rel-RMS of float RGB on the stride-4 grid, by DDIM steps
BF16_BAR_U8_CHAINED = {20: 1.0, 50: 0.95}           # mean |u8 - exact image| in LSB
MX_BAR_RGB_CHAINED = {0: 8.3e-2, 1: 1.23e-1}        # by fp8_linear
MX_BAR_U8_CHAINED = {0: 3.2, 1: 4.6}


def _chained_image_check(sd, got_latents, ref_rgb64_s4, label):
    """decode the GPU's own final latents (float RGB + truncating u8 image through the C ABI) and compare with the fp64 oracle's image of ITS fp64 latent on the
    stride-4 grid the fixtures hold: returns [(sample, rel-RMS of float RGB, mean |u8 - exact| in LSB)]"""
    idx = sorted(ref_rgb64_s4)
    own = np.stack([got_latents[i] for i in idx]).astype(np.float32)
    rgb = sd.autoencoder.decode_latent(own * np.float32(1.0 / 0.18215))
    img = sd.latent_to_image(own)
    out = []
    for k, i in enumerate(idx):
        r = _rel_rms(rgb[k][:, ::4, ::4], ref_rgb64_s4[i])
        exact = np.clip((ref_rgb64_s4[i] + 1.0) * 127.5, 0, 255).transpose(1, 2, 0)
        d = float(np.abs(img[k].astype(np.float64)[::4, ::4] - exact).mean())
        print(f"{label}, sample {i}: decode of the GPU's OWN latent vs the fp64 image of the fp64 latent: rel-RMS RGB = {r:.3e}, u8 mean |d| = {d:.2f} LSB")
        out.append((i, r, d))
    return out


def _rel_rms(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))


@pytest.fixture(scope="module")
def sd_full():
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig())
    sd.load_weights(syn.SyntheticWeights())
    yield sd
    sd.close()


def _inputs():
    return syn.initial_latent(0)[None], syn.cond_context(0)[None], syn.uncond_context()


def test_unet_forward_full(sd_full):
    """One UNet::forward (unet/mod.rs:109-143) at t = 999 and t = 49, T = 77."""
    g = np.load(GOLD / "sd14_synth_unet.npz")
    lat, ctx, _ = _inputs()
    for t in (999, 49):
        got = sd_full.unet.forward(lat, [t], ctx)[0].astype(np.float64)
        e32 = np.abs(got - g[f"eps32_t{t}"]).max()
        e64 = np.abs(got - g[f"eps64_t{t}"]).max()
        ref_gap = np.abs(g[f"eps32_t{t}"].astype(np.float64) - g[f"eps64_t{t}"]).max()
        print(f"unet t={t}: |gpu-f32|={e32:.2e} |gpu-f64|={e64:.2e} |f32-f64|={ref_gap:.2e}")
        assert e64 < 2e-5, f"t={t}: {e64}"


def test_unet_forward_full_batch4(sd_full):
    """batch 4 (M = 16384 ... 256 rows: GEMM shapes outside the measured batch-1 tile table, chosen by the cost model incl. the
    large-tile kernel): every sample equals the batch-1 golden vector."""
    g = np.load(GOLD / "sd14_synth_unet.npz")
    lat, ctx, _ = _inputs()
    got = sd_full.unet.forward(np.repeat(lat, 4, axis=0), [999], np.repeat(ctx, 4, axis=0)).astype(np.float64)
    for i in range(4):
        assert np.abs(got[i] - g["eps64_t999"]).max() < 2e-5, i
    assert np.array_equal(got[0], got[3])   # position in the batch does not matter


def test_config1_one_step(sd_full):
    """configs[0]: 1 DDIM step, guidance 1.0 ("CFG off" still runs both forwards, Q: forward_diffuser)."""
    g = np.load(GOLD / "sd14_synth_cfg1.npz")
    lat, ctx, unc = _inputs()
    got = sd_full.sample_latent(ctx, unc, 1.0, 1, init_latent=lat)[0].astype(np.float64)
    assert np.abs(got - g["latent32"]).max() < 1e-3
    assert np.abs(got - g["latent64"]).max() < 1e-3
    img = sd_full.latent_to_image(g["latent32"][None])[0]
    assert np.abs(img.astype(np.int16) - g["rgb_u8"].astype(np.int16)).max() <= 1


def test_config2_20_steps_cfg(sd_full):
    """configs[1]: 20-step DDIM, CFG 7.5, batch 1 fp32 -- the benchmarked configuration."""
    g = np.load(GOLD / "sd14_synth_cfg2.npz")
    lat, ctx, unc = _inputs()
    got = sd_full.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)[0].astype(np.float64)
    ref32 = g["latents32"][-1].astype(np.float64)
    ref64 = g["latent64"]
    e32, e64 = np.abs(got - ref32).max(), np.abs(got - ref64).max()
    gap = float(g["step_err"][-1])
    print(f"final latent: |gpu-f32|={e32:.2e} |gpu-f64|={e64:.2e} |f32-f64|={gap:.2e} absmax={np.abs(ref64).max():.1f}")
    assert e32 < 1e-3 and e64 <= max(1e-3, 2 * gap)

    # decoded float RGB on the stride-4 grid + u8 image, from the GPU's own latent (whole sample_image path)
    rgb = sd_full.autoencoder.decode_latent((got[None] * (1.0 / 0.18215)).astype(np.float32))[0]
    d32 = np.abs(rgb[:, ::4, ::4] - g["rgb32_s4"]).max()
    d64 = np.abs(rgb[:, ::4, ::4].astype(np.float64) - g["rgb64_s4"]).max()
    print(f"float RGB (stride-4 grid): |gpu-f32|={d32:.2e} |gpu-f64|={d64:.2e}")
    assert d32 < 1e-3 and d64 < 1e-3
    st = g["rgb32_stats"]
    assert np.abs(rgb.mean(axis=(1, 2)) - st[0]).max() < 1e-4 and np.abs(rgb.std(axis=(1, 2)) - st[1]).max() < 1e-4
    # round 6: every pixel of the float image, not only the stride-4 grid (fixture: the fp32 oracle's decode in 2^-13 fixed point, tests/golden/gen_golden_rgb_full.py)
    full = np.load(GOLD / "sd14_synth_cfg2_rgb_full.npz")
    ref_full = full["rgb32_q"].astype(np.float64) / float(1 << int(full["shift"]))
    dfull = np.abs(rgb.astype(np.float64) - ref_full).max()
    print(f"float RGB (all 512 x 512 x 3 values): |gpu-f32| = {dfull:.2e} (fixture quantisation 6.1e-5)")
    assert rgb.shape == ref_full.shape and dfull < 1e-3

    img = sd_full.sample_image(ctx, unc, 7.5, 20, init_latent=lat)[0]
    diff = np.abs(img.astype(np.int16) - g["rgb_u8"].astype(np.int16))
    print(f"u8 image: max diff {diff.max()} LSB, {np.count_nonzero(diff)} of {diff.size} bytes differ")
    assert diff.max() <= 1


# ---- the reference's real prompt shape: unpadded contexts, Tc != Tu (stablediffusion/mod.rs:198-210) ----------------------
def _q2():
    path = GOLD / "sd14_synth_q2.npz"
    if not path.exists():
        pytest.skip("tests/golden/sd14_synth_q2.npz not generated yet (tests/golden/gen_golden_q2.py)")
    return np.load(path), syn.initial_latent(0)[None], syn.cond_context(0, 77)[None], syn.uncond_context(2)


def test_unpadded_contexts_full_size_fp32(sd_full):
    """Tc = 77, Tu = 2 at the full model size, 20 steps, CFG 7.5, fp32: the two halves of the CFG batch attend over different key counts
    (per-row key count of the batch-2n forward).  Same bars as configs[1]."""
    g, lat, ctx, unc = _q2()
    got = sd_full.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)[0].astype(np.float64)
    e32, e64 = np.abs(got - g["latent32"]).max(), np.abs(got - g["latent64"]).max()
    gap = np.abs(g["latent32"].astype(np.float64) - g["latent64"]).max()
    print(f"Tc=77 Tu=2 final latent: |gpu-f32|={e32:.2e} |gpu-f64|={e64:.2e} |f32-f64|={gap:.2e} absmax={np.abs(g['latent64']).max():.1f}")
    assert e32 < 1e-3 and e64 <= max(1e-3, 2 * gap)


@pytest.mark.parametrize("precision", [1, 2])
def test_unpadded_contexts_full_size_reduced_precision(precision):
    """the same call in bf16 and in precision 2 (defaults): relative RMS of the final latent against the fp64 fixture under the bars of the
    T = Tu = 77 cases (bf16: BF16_BAR_LATENT20; precision 2: MX_BAR_LATENT20 of the default set)"""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    g, lat, ctx, unc = _q2()
    sd = StableDiffusion(ModelConfig(precision=precision))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        got = sd.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)[0]
        r = _rel_rms(got, g["latent64"])
        print(f"Tc=77 Tu=2, precision {precision}: rel-RMS of the 20-step latent vs fp64 = {r:.3e}")
        assert np.isfinite(got).all() and r < (BF16_BAR_LATENT20 if precision == 1 else MX_BAR_LATENT20[MX_DEFAULT_WIDE])
    finally:
        sd.close()


def test_per_step_drift(sd_full):
    """Latent after each of the first 3 steps stays within the oracle's own f32/f64 gap (x4)."""
    g = np.load(GOLD / "sd14_synth_cfg2.npz")
    lat, ctx, unc = _inputs()
    # n_steps=20 schedule truncated is not expressible through the reference surface; instead check
    # the 1-step-of-20 latent by running the same schedule with guidance and comparing step 0 only
    # via the UNet forwards at t=999 (cond / uncond) recombined on the host in f64.
    eu = sd_full.unet.forward(lat, [999], unc[None])[0].astype(np.float64)
    ec = sd_full.unet.forward(lat, [999], ctx)[0].astype(np.float64)
    a = syn.alphas_cumprod().astype(np.float64)
    e = eu + (ec - eu) * 7.5
    x0 = (lat[0] - e * np.sqrt(1 - a[999])) / np.sqrt(a[999])
    x1 = x0 * np.sqrt(a[949]) + e * np.sqrt(1 - a[949])
    err = np.abs(x1 - g["latents32"][0]).max()
    print(f"step-0 latent: |gpu-f32 oracle|={err:.2e}  (oracle f32/f64 gap {g['step_err'][0]:.2e})")
    assert err < max(1e-4, 4 * float(g["step_err"][0]))


def test_bf16_full_size_against_the_fp64_fixtures():
    """precision = 1 at the full SD v1.4 size with the shipped bf16 tile table (large-tile LDS-DMA GEMMs, bf16 matrix-core
    attention): relative RMS against the fp64 golden vectors -- one UNet forward, the 20-step CFG latent, decoded RGB.
    Bars as in test_bf16_gpu.py ("expect ~1e-2 bf16")."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(precision=1))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        lat, ctx, unc = _inputs()

        def rel_rms(a, b):
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))

        gu = np.load(GOLD / "sd14_synth_unet.npz")
        r_unet = rel_rms(sd.unet.forward(lat, [999], ctx)[0], gu["eps64_t999"])
        g2 = np.load(GOLD / "sd14_synth_cfg2.npz")
        got = sd.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)[0]
        r_lat = rel_rms(got, g2["latent64"])
        rgb = sd.autoencoder.decode_latent((g2["latent64"][None] * (1.0 / 0.18215)).astype(np.float32))[0]
        r_rgb = rel_rms(rgb[:, ::4, ::4], g2["rgb64_s4"])
        print(f"bf16 full size: rel-RMS UNet forward {r_unet:.3e}, 20-step CFG latent {r_lat:.3e}, decoded RGB {r_rgb:.3e}")
        assert np.isfinite(got).all() and r_unet < BF16_BAR_UNET and r_lat < BF16_BAR_LATENT20 and r_rgb < BF16_BAR_RGB
    finally:
        sd.close()


def test_bf16_kernel_row_tiles_in_the_model_are_bit_identical(tmp_path):
    """option conv3_reuse (default 1): the 3x3 convolutions that choose the 256 x 320 / 256 x 256 tile run on k_gemm_bf16t.hip (one staged activation tile per
    kernel row).  Same products in the same order: one batch-32 UNet forward (the batch of BASELINE.json configs[2]'s CFG step) is bit-identical either way, and the
    choice dump shows the tiles were taken -- concat inputs, residual / time-embedding epilogues included."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(precision=1))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        n = 32
        lat = np.stack([syn.initial_latent(i % 4) for i in range(n)])
        ctx = np.stack([syn.cond_context(i % 4) for i in range(n)])
        sd.set_option("record_shapes", 1)
        on = sd.unet.forward(lat, [500], ctx)
        sd.set_option("dump_choices", str(tmp_path / "choices.txt"))
        sd.set_option("record_shapes", 0)
        taken = [ln for ln in (tmp_path / "choices.txt").read_text().splitlines() if "cfg=104" in ln or "cfg=105" in ln]
        sd.set_option("conv3_reuse", 0)
        off = sd.unet.forward(lat, [500], ctx)
        print(f"kernel-row tiles: {len(taken)} GEMM shapes of the batch-{n} forward")
        assert len(taken) >= 6 and np.isfinite(on).all()
        np.testing.assert_array_equal(on, off)
    finally:
        sd.close()


# ---- BASELINE.json configs[2]: 50 steps, batch 16, bf16 ------------------------------------------------------------
def _cfg3_inputs(n):
    lat = np.stack([syn.initial_latent(i) for i in range(n)])
    ctx = np.repeat(syn.cond_context(0)[None], n, axis=0)      # one prompt for the whole batch, like bench.py
    return lat, ctx, syn.uncond_context()


def test_config3_50_steps_fp32(sd_full):
    """The 50-step schedule t = 999, 979, .., 19 (stablediffusion/mod.rs:111,123) in fp32 against the fp64 fixture
    (tests/golden/gen_golden_cfg3.py), batch 2 = the first two samples of configs[2]."""
    g = np.load(GOLD / "sd14_synth_cfg3.npz")
    lat, ctx, unc = _cfg3_inputs(2)
    got = sd_full.sample_latent(ctx, unc, 7.5, 50, init_latent=lat).astype(np.float64)
    for i in range(2):
        e = np.abs(got[i] - g["latent64"][i]).max()
        print(f"50-step fp32, sample {i}: |gpu-f64| = {e:.2e} (absmax {np.abs(g['latent64'][i]).max():.1f})")
        assert e < 1e-3


def _more():
    """samples 7 and 15 of the batched configurations (tests/golden/gen_golden_more_samples.py): index -> position in the fixture's arrays"""
    m = np.load(GOLD / "sd14_synth_more.npz")
    return m, {int(i): j for j, i in enumerate(m["index"].tolist())}


def test_config3_bf16_batch16_50_steps():
    """configs[2] as BASELINE.json states it: batch 16, 50 DDIM steps, CFG 7.5, bf16 on one GPU.  Samples 0, 1, 7 and 15 (round 5: four, not two) are
    compared with the fp64 oracle's batch-1 runs (the reference defines batch > 1 as independent samples);
    the bars are 1.5x the values measured on MI355X (printed).  Every sample must be finite and must not depend on the
    batch position: sample 0's latent is reproduced bit-exactly when it is also placed at position 14."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    g = np.load(GOLD / "sd14_synth_cfg3.npz")
    m, at = _more()
    ref = {0: g["latent64"][0], 1: g["latent64"][1], 7: m["latent64_s50"][at[7]], 15: m["latent64_s50"][at[15]]}
    ref_rgb = {0: g["rgb64_s4"][0], 1: g["rgb64_s4"][1], 7: m["rgb64_s50_s4"][at[7]], 15: m["rgb64_s50_s4"][at[15]]}
    sd = StableDiffusion(ModelConfig(precision=1))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        lat, ctx, unc = _cfg3_inputs(16)
        lat[14] = lat[0]
        got = sd.sample_latent(ctx, unc, 7.5, 50, init_latent=lat)
        assert np.isfinite(got).all()
        assert np.array_equal(got[0], got[14])
        for i in sorted(ref):
            r = _rel_rms(got[i], ref[i])
            print(f"bf16 B=16 S=50, sample {i}: rel-RMS of the final latent vs fp64 = {r:.3e}")
            assert r < BF16_BAR_LATENT50
        rgb = sd.autoencoder.decode_latent((np.stack([ref[i] for i in sorted(ref)]) * (1.0 / 0.18215)).astype(np.float32))
        for k, i in enumerate(sorted(ref)):
            r = _rel_rms(rgb[k][:, ::4, ::4], ref_rgb[i])
            print(f"bf16 decode of the fp64 latent, sample {i}: rel-RMS RGB = {r:.3e}")
            assert r < BF16_BAR_RGB
        for i, r, d in _chained_image_check(sd, got, ref_rgb, "bf16 B=16 S=50"):
            assert r < BF16_BAR_RGB_CHAINED[50] and d < BF16_BAR_U8_CHAINED[50], (i, r, d)
    finally:
        sd.close()


# ---- configs[3]: bf16, 64 images over 8 GPUs = batch 8 per GPU, 20 steps: the per-GPU shard, its own golden test (round 5) ---------------
def test_config4_shard_bf16_batch8_20_steps():
    """The shard one GPU of BASELINE.json configs[3] runs (and `bench.py --config 3` times): batch 8, 20 DDIM steps, CFG 7.5, bf16.  Samples 0, 1 (fixture
    sd14_synth_cfg5.npz, exact network) and 7 -- the shard's last image -- (sd14_synth_more.npz) against the fp64 oracle's batch-1 runs; the u8 image of sample 7
    against the oracle's decode of ITS latent; every sample finite and position-independent (sample 0 again at position 6, bit-exact)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    g = np.load(GOLD / "sd14_synth_cfg5.npz")
    m, at = _more()
    ref = {0: g["latent64"][0], 1: g["latent64"][1], 7: m["latent64_s20"][at[7]]}
    sd = StableDiffusion(ModelConfig(precision=1))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        lat, ctx, unc = _cfg3_inputs(8)
        lat[6] = lat[0]
        got = sd.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)
        assert np.isfinite(got).all()
        assert np.array_equal(got[0], got[6])
        for i in sorted(ref):
            r = _rel_rms(got[i], ref[i])
            print(f"bf16 B=8 S=20, sample {i}: rel-RMS of the final latent vs fp64 = {r:.3e}")
            assert r < BF16_BAR_LATENT20
        rgb = sd.autoencoder.decode_latent((ref[7][None] * (1.0 / 0.18215)).astype(np.float32))[0]
        r = _rel_rms(rgb[:, ::4, ::4], m["rgb64_s20_s4"][at[7]])
        print(f"bf16 decode of the fp64 latent, sample 7: rel-RMS RGB = {r:.3e}")
        assert r < BF16_BAR_RGB
        img = sd.latent_to_image(ref[7][None].astype(np.float32))[0].astype(np.float64)      # truncating u8 of the bf16 decode: within a few LSB of the exact decode on the grid
        exact = np.clip((m["rgb64_s20_s4"][at[7]] + 1.0) * 127.5, 0, 255).transpose(1, 2, 0)
        d = np.abs(img[::4, ::4] - exact)
        print(f"bf16 u8 image of sample 7 vs the exact decode: mean |d| = {d.mean():.2f} LSB, max {d.max():.1f}")
        assert d.mean() < 2.5
        ref_rgb = {0: g["rgb64_s4"][0], 1: g["rgb64_s4"][1], 7: m["rgb64_s20_s4"][at[7]]}
        for i, r, dd in _chained_image_check(sd, got, ref_rgb, "bf16 B=8 S=20"):
            assert r < BF16_BAR_RGB_CHAINED[20] and dd < BF16_BAR_U8_CHAINED[20], (i, r, dd)
    finally:
        sd.close()


# ---- configs[4]: precision = 2 at FULL size, batch 16 (the per-GPU shard of 128 images over 8 GPUs), 20 steps ---------------------------
# Bars = 1.5 x the relative RMS measured on MI355X (round 3, tools/probes/r03m_dump.py + r03m_compare.py; profiles/README.md): what MXFP8 costs
# over 20 chained CFG steps and a decode at the real model size -- no longer an 8x8-latent statement.
#   measured, samples 0 / 1:   fp8_linear = 1 (default)   latent vs exact 8.06e-2 / 7.90e-2
#                              fp8_linear = 0             latent vs exact 5.16e-2 / 5.14e-2, vs the fp64 network with the same quantisation 6.49e-2 / 6.41e-2
#                              (the quantisation of fp8_linear = 0 alone, fp64 vs fp64: 5.09e-2 / 5.08e-2 -- the GPU pays what the format costs)
#                              decode of the exact latent (both modes: the decoder's fp8 set does not depend on fp8_linear): RGB 2.05e-2
MX_BAR_LATENT20 = {0: 7.8e-2, 1: 1.21e-1}      # vs the exact fp64 network, by fp8_linear
MX_DEFAULT_WIDE = 0                             # the engine's default fp8_linear (accuracy budget: 6e-2 on this latent; measured 5.2e-2 / 8.1e-2)
MX_BAR_LATENT20_SAMEQ = 9.8e-2                  # fp8_linear = 0 vs the fp64 network with the same quantisation
MX_BAR_RGB = {0: 3.1e-2, 1: 3.1e-2}


@pytest.mark.parametrize("wide", [1, 0])
def test_config5_mxfp8_batch16_20_steps(wide):
    """configs[4] as BASELINE.json states it for one GPU: batch 16, 20 DDIM steps, CFG 7.5, precision = 2 (reference arithmetic:
    stablediffusion/mod.rs:102-160 in f32).  wide = 0: the default -- bf16 + MXFP8 on the ResBlock / ResnetBlock 3x3 convolutions; wide = 1: option fp8_linear = 1, also the transformer blocks'
    Linear layers and the UNet's 1x1 / up / down convolutions.  Samples 0, 1, 7 and 15
    against the fp64 fixtures of tests/golden/gen_golden_cfg5.py / gen_golden_more_samples.py: the exact network (what the format costs end to end, 20 chained CFG steps
    at the real model size) and -- for wide = 0, whose quantisation the fixture reproduces -- the fp64 network with the same MXFP8
    quantisation.  Every sample finite; a sample's result does not depend on its batch position (bit-exact)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    path = GOLD / "sd14_synth_cfg5.npz"
    if not path.exists():
        pytest.skip("tests/golden/sd14_synth_cfg5.npz not generated yet (tests/golden/gen_golden_cfg5.py)")
    g = np.load(path)
    sd = StableDiffusion(ModelConfig(precision=2))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        sd.set_option("fp8_linear", wide)
        m, at = _more()
        exact = {0: g["latent64"][0], 1: g["latent64"][1], 7: m["latent64_s20"][at[7]], 15: m["latent64_s20"][at[15]]}
        sameq = {0: g["latent64_mx"][0], 1: g["latent64_mx"][1], 7: m["latent64_mx_s20"][at[7]], 15: m["latent64_mx_s20"][at[15]]}
        lat, ctx, unc = _cfg3_inputs(16)
        lat[14] = lat[0]
        got = sd.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)
        assert np.isfinite(got).all()
        assert np.array_equal(got[0], got[14])
        for i in sorted(exact):      # round 5: samples 0, 1, 7 and 15
            r_exact, r_same, fmt = _rel_rms(got[i], exact[i]), _rel_rms(got[i], sameq[i]), _rel_rms(sameq[i], exact[i])
            print(f"precision 2 (fp8_linear={wide}), B=16 S=20, sample {i}: rel-RMS of the final latent vs exact fp64 = {r_exact:.3e}, vs fp64 with the "
                  f"ResBlock-conv MXFP8 quantisation = {r_same:.3e} (that quantisation alone, quantised fp64 vs exact fp64: {fmt:.3e})")
            assert r_exact < MX_BAR_LATENT20[wide]
            if not wide:
                assert r_same < MX_BAR_LATENT20_SAMEQ
        rgb = sd.autoencoder.decode_latent((g["latent64"][:2] * (1.0 / 0.18215)).astype(np.float32))
        for i in range(2):
            r = _rel_rms(rgb[i][:, ::4, ::4], g["rgb64_s4"][i])
            print(f"precision 2 (fp8_linear={wide}) decode of the exact fp64 latent, sample {i}: rel-RMS RGB = {r:.3e}")
            assert r < MX_BAR_RGB[wide]
        ref_rgb = {0: g["rgb64_s4"][0], 1: g["rgb64_s4"][1], 7: m["rgb64_s20_s4"][at[7]], 15: m["rgb64_s20_s4"][at[15]]}
        for i, r, d in _chained_image_check(sd, got, ref_rgb, f"precision 2 (fp8_linear={wide}) B=16 S=20"):
            assert r < MX_BAR_RGB_CHAINED[wide] and d < MX_BAR_U8_CHAINED[wide], (i, r, d)
    finally:
        sd.close()


def test_precision2_runs_mxfp8_on_exactly_the_resblock_convolutions(tmp_path):
    """Which layers precision = 2 puts on MXFP8 operands, pinned by the engine's choice dump (round 5: "fp8 conv" must not shrink silently): one batch-16 UNet
    forward = the 44 ResBlock 3x3 convolutions (unet/mod.rs:716,729; every level has >= fp8_min_rows output rows at this batch) and nothing else; one decode = the
    28 ResnetBlock 3x3 convolutions (autoencoder/mod.rs:516-523); with fp8_linear=1 the transformer blocks' Linear layers and the 1x1 / up / down convolutions join."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(precision=2))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
        lat, ctx, _ = _cfg3_inputs(16)

        def fp8_launches(fn, name):
            sd.set_option("record_shapes", 1)
            fn()
            sd.set_option("dump_choices", str(tmp_path / name))
            sd.set_option("record_shapes", 0)
            rows = [ln.split() for ln in (tmp_path / name).read_text().splitlines()]
            fp8 = [(r[0], r[1], int(r[-1][1:])) for r in rows if "fp8" in r]
            rest = [(r[0], r[1], int(r[-1][1:])) for r in rows if "fp8" not in r]
            return fp8, rest

        fp8, rest = fp8_launches(lambda: sd.unet.forward(lat, [500], ctx), "unet.txt")
        print(f"precision 2 UNet forward: {sum(c for *_, c in fp8)} MXFP8 launches over {len(fp8)} shapes, {sum(c for *_, c in rest)} bf16 / fp32 GEMM launches")
        assert sum(c for *_, c in fp8) == 44 and all(k == "k3" for _, k, _ in fp8)
        fp8d, _ = fp8_launches(lambda: sd.autoencoder.decode_latent(lat[:1]), "dec.txt")
        print(f"precision 2 decode: {sum(c for *_, c in fp8d)} MXFP8 launches")
        assert sum(c for *_, c in fp8d) == 28 and all(k == "k3" for _, k, _ in fp8d)
        sd.set_option("fp8_linear", 1)
        fp8w, _ = fp8_launches(lambda: sd.unet.forward(lat, [500], ctx), "unet_wide.txt")
        print(f"precision 2, fp8_linear=1, UNet forward: {sum(c for *_, c in fp8w)} MXFP8 launches")
        assert sum(c for *_, c in fp8w) > 44 + 100 and any(k == "k1" for _, k, _ in fp8w)
    finally:
        sd.close()
