"""Model-level parity of the HIP path (through the C ABI) against the CPU oracle.

Runs the half-width model (same topology, same head dims 40/80/160, 16x16
latent) so the oracle finishes in seconds; the full-size configuration is
checked against committed golden fixtures in test_golden_gpu.py.

Tolerances (BASELINE.json north_star: max per-pixel |delta| < 1e-3 vs the fp32
reference): absolute 1e-3 on latents / float RGB, with the fp64
oracle as tie-breaker -- |gpu - f64| <= max(1e-3, 2*|cpu_f32 - f64|) -- and
<= 1 LSB on the u8 image (the reference truncates, so a +-1 flip is expected
wherever |delta|*127.5 crosses an integer).
"""
import numpy as np
import pytest
import torch

from oracle import sd_oracle as O
from stable_diffusion_burn_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _inputs(d, n, T, Tu):
    lat = np.stack([syn.initial_latent(i, d.latent_h, d.latent_w) for i in range(n)])
    ctx = np.stack([syn.cond_context(i, T, d.ctx_dim) for i in range(n)])
    unc = syn.uncond_context(Tu, d.ctx_dim)
    return lat, ctx, unc


def _oracles(synth, d):
    a = syn.alphas_cumprod()
    return (O.StableDiffusionOracle(synth, a, d, torch.float32), O.StableDiffusionOracle(synth, a, d, torch.float64))


def _assert_close(got, ref32, ref64, what, atol=1e-3):
    got = np.asarray(got, np.float64)
    r32 = np.asarray(ref32, np.float64)
    r64 = np.asarray(ref64, np.float64)
    assert np.isfinite(got).all(), f"{what}: non-finite"
    e64 = np.abs(got - r64).max()
    e32 = np.abs(r32 - r64).max()
    bound = max(atol, 2 * e32)
    assert e64 <= bound, f"{what}: max|gpu-f64|={e64:.3e} > {bound:.3e} (|f32-f64|={e32:.3e}, |gpu-f32|={np.abs(got - r32).max():.3e})"
    return e64, e32


@pytest.mark.parametrize("t", [999, 49])
def test_unet_forward(sd_tiny, synth, tiny_dims, t):
    """UNet::forward (unet/mod.rs:109-143), batch 2 with an unpadded context (T = 7)."""
    d = tiny_dims
    lat, ctx, _ = _inputs(d, 2, 7, 2)
    o32, o64 = _oracles(synth, d)
    got = sd_tiny.unet.forward(lat, [t], ctx)
    r32 = o32.unet.forward(torch.from_numpy(lat), t, torch.from_numpy(ctx)).numpy()
    r64 = o64.unet.forward(torch.from_numpy(lat), t, torch.from_numpy(ctx)).numpy()
    e64, e32 = _assert_close(got, r32, r64, f"unet_forward t={t}", atol=1e-4)
    print(f"unet t={t}: |gpu-f64|={e64:.2e} |f32-f64|={e32:.2e}")


@pytest.mark.parametrize("tile", [100, 103, 200, 203, 204, 205])
def test_unet_forward_large_tiles_forced(sd_tiny, synth, tiny_dims, tile):
    """every eligible GEMM of the UNet on one k_gemm2x.hip (100+) / k_gemm3x.hip (200+) tile: its residual / time-embedding /
    split-K epilogues at model level."""
    d = tiny_dims
    lat, ctx, _ = _inputs(d, 2, 7, 2)
    o32, o64 = _oracles(synth, d)
    try:
        sd_tiny.set_option("gemm_tile", tile)
        got = sd_tiny.unet.forward(lat, [500], ctx)
    finally:
        sd_tiny.set_option("gemm_tile", "auto")
    r32 = o32.unet.forward(torch.from_numpy(lat), 500, torch.from_numpy(ctx)).numpy()
    r64 = o64.unet.forward(torch.from_numpy(lat), 500, torch.from_numpy(ctx)).numpy()
    _assert_close(got, r32, r64, f"unet_forward tile={tile}", atol=1e-4)


def test_unet_forward_fp32_matrix_instruction_only(sd_tiny, synth, tiny_dims):
    """gemm_f32s=0 / attn_split=0: every product on v_mfma_f32_16x16x4_f32 (round 1's arithmetic).  Same bar; and the default
    path (fp32 operands as three bf16 terms, six partial products) agrees with it to fp32 rounding noise."""
    d = tiny_dims
    lat, ctx, _ = _inputs(d, 2, 7, 2)
    o32, o64 = _oracles(synth, d)
    split = sd_tiny.unet.forward(lat, [700], ctx)
    try:
        sd_tiny.set_option("gemm_f32s", 0)
        sd_tiny.set_option("attn_split", 0)
        plain = sd_tiny.unet.forward(lat, [700], ctx)
    finally:
        sd_tiny.set_option("gemm_f32s", 1)
        sd_tiny.set_option("attn_split", 1)
    r32 = o32.unet.forward(torch.from_numpy(lat), 700, torch.from_numpy(ctx)).numpy()
    r64 = o64.unet.forward(torch.from_numpy(lat), 700, torch.from_numpy(ctx)).numpy()
    e_plain, _ = _assert_close(plain, r32, r64, "unet_forward fp32 MFMA only", atol=1e-4)
    e_split, e32 = _assert_close(split, r32, r64, "unet_forward split kernels", atol=1e-4)
    print(f"|gpu-f64|: split kernels {e_split:.2e}, fp32 matrix instruction {e_plain:.2e}; the fp32 oracle itself {e32:.2e}")
    assert e_split <= 2.0 * max(e_plain, e32)
    assert np.abs(split - plain).max() <= 2e-5 * max(1.0, np.abs(r64).max())


def test_unet_forward_batch_independent(sd_tiny, tiny_dims):
    """Per-sample results do not depend on the batch (batch = n independent samples)."""
    d = tiny_dims
    lat, ctx, _ = _inputs(d, 2, 5, 2)
    both = sd_tiny.unet.forward(lat, [500], ctx)
    one = sd_tiny.unet.forward(lat[1:2], [500], ctx[1:2])
    assert np.abs(both[1:2] - one).max() <= 1e-5


# (3, ..): 1000 / 3 = 333 -> t = 999, 666, 333, 0: FOUR iterations (quirk Q5, step_by)
@pytest.mark.parametrize("n_steps,scale,T,Tu", [(1, 1.0, 7, 7), (4, 7.5, 7, 2), (5, 7.5, 3, 6), (3, 7.5, 7, 2)])
def test_sample_latent(sd_tiny, synth, tiny_dims, n_steps, scale, T, Tu):
    """sample_latent (stablediffusion/mod.rs:102-160): DDIM + CFG, Tc != Tu, config-1 style 1 step."""
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, T, Tu)
    o32, o64 = _oracles(synth, d)
    got = sd_tiny.sample_latent(ctx, unc, scale, n_steps, init_latent=lat)
    args = (torch.from_numpy(ctx), torch.from_numpy(unc), scale, n_steps, torch.from_numpy(lat))
    r32 = o32.sample_latent(*args).numpy()
    r64 = o64.sample_latent(*args).numpy()
    e64, e32 = _assert_close(got, r32, r64, f"sample_latent steps={n_steps}")
    print(f"sample_latent steps={n_steps}: |gpu-f64|={e64:.2e} |f32-f64|={e32:.2e} absmax={np.abs(r64).max():.1f}")


def test_sample_latent_batch2_matches_single(sd_tiny, tiny_dims):
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 2, 7, 2)
    both = sd_tiny.sample_latent(ctx, unc, 7.5, 3, init_latent=lat)
    for i in range(2):
        one = sd_tiny.sample_latent(ctx[i:i + 1], unc, 7.5, 3, init_latent=lat[i:i + 1])
        scale = max(1.0, np.abs(one).max())
        assert np.abs(both[i:i + 1] - one).max() <= 2e-5 * scale, f"sample {i}"


@pytest.mark.parametrize("precision", [0, 1, 2])
def test_shared_cfg_prefix_equals_two_full_forwards(synth, tiny_dims, precision):
    """Round 5, option cfg_share (default 1): the part of the UNet in front of the first cross attention (input block 1's ResBlock and its transformer up to the
    cross attention) is computed ONCE for the two halves of a CFG step -- they are the reference's two forwards of the SAME x and t (stablediffusion/mod.rs:173-179)
    and differ only in the context -- and duplicated there.  cfg_share=0 computes both halves.  Same values per sample; the GEMMs of the prefix run at half the rows, so
    their tile / split-K choice (hence the summation order) may differ: fp32 to 2e-5 relative, reduced precisions to their operator bars over 3 chained steps."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    # fp32: the half-width model; bf16 / MXFP8 need channel counts that are multiples of 64: the full-width model at an 8x8 latent (as tests/test_bf16_gpu.py)
    d = tiny_dims if precision == 0 else O.Dims(320, 8, 768, 8, 8, 64)
    sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch, precision=precision))
    try:
        sd.load_weights(synth, clip=False, vae_encoder=False)
        lat, ctx, unc = _inputs(d, 3, 7, 2)
        shared = sd.sample_latent(ctx, unc, 7.5, 3, init_latent=lat)
        ks = sd.last_call_stats()["kernels"]
        sd.set_option("cfg_share", 0)
        full = sd.sample_latent(ctx, unc, 7.5, 3, init_latent=lat)
        kf = sd.last_call_stats()["kernels"]
        scale = max(1.0, np.abs(full).max())
        err = np.abs(shared - full).max() / scale
        print(f"precision {precision}: shared prefix vs two full forwards, 3 steps: max|d| / absmax = {err:.2e}; kernels {ks} vs {kf}")
        assert np.isfinite(shared).all() and err <= (2e-5 if precision == 0 else 2e-2)
    finally:
        sd.close()


def test_decode_latent(sd_tiny, synth, tiny_dims):
    """Autoencoder::decode_latent (autoencoder/mod.rs:68-71, 205-217)."""
    d = tiny_dims
    z = (np.stack([syn.initial_latent(10 + i, d.latent_h, d.latent_w) for i in range(2)]) * 3.0).astype(np.float32)
    o32, o64 = _oracles(synth, d)
    got = sd_tiny.autoencoder.decode_latent(z)
    r32 = o32.decoder.decode_latent(torch.from_numpy(z)).numpy()
    r64 = o64.decoder.decode_latent(torch.from_numpy(z)).numpy()
    e64, e32 = _assert_close(got, r32, r64, "decode_latent")
    print(f"decode: |gpu-f64|={e64:.2e} |f32-f64|={e32:.2e} absmax={np.abs(r64).max():.1f}")


def test_latent_to_image_and_sample_image(sd_tiny, synth, tiny_dims):
    """latent_to_image (stablediffusion/mod.rs:69-100) and the whole sample_image (:51-67): u8 within 1 LSB."""
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    o32, _ = _oracles(synth, d)
    z = (syn.initial_latent(3, d.latent_h, d.latent_w)[None] * 0.5).astype(np.float32)
    img = sd_tiny.latent_to_image(z)
    ref, _ = o32.latent_to_image(torch.from_numpy(z))
    assert img.shape == ref.shape and img.dtype == np.uint8
    assert np.abs(img.astype(np.int16) - ref.astype(np.int16)).max() <= 1
    # not a constant image
    assert img.std() > 1.0

    full = sd_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
    ref_full = o32.sample_image(torch.from_numpy(ctx), torch.from_numpy(unc), 7.5, 2, torch.from_numpy(lat))
    diff = np.abs(full.astype(np.int16) - ref_full.astype(np.int16))
    assert diff.max() <= 1, f"u8 image differs by {diff.max()} LSB at {np.count_nonzero(diff > 1)} pixels"


def test_sample_image_seed_path(sd_tiny, tiny_dims):
    """init_latent = NULL draws x_T on the device from `seed` (deterministic, seed-dependent)."""
    d = tiny_dims
    _, ctx, unc = _inputs(d, 1, 7, 2)
    a = sd_tiny.sample_latent(ctx, unc, 7.5, 1, init_latent=None, seed=5)
    b = sd_tiny.sample_latent(ctx, unc, 7.5, 1, init_latent=None, seed=5)
    c = sd_tiny.sample_latent(ctx, unc, 7.5, 1, init_latent=None, seed=6)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and np.isfinite(a).all()


def test_run_to_run_bit_reproducible(sd_tiny, tiny_dims):
    """No atomics in the results path: identical bits on repeated calls."""
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    a = sd_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
    b = sd_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
    assert np.array_equal(a, b)


def test_shape_errors_raise(sd_tiny, tiny_dims):
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    with pytest.raises(ValueError):
        sd_tiny.sample_latent(ctx[:, :, :-1], unc, 7.5, 1, init_latent=lat)
    with pytest.raises(ValueError):
        sd_tiny.latent_to_image(lat[:, :3])
    from stable_diffusion_burn_amd import SdmiError
    with pytest.raises(SdmiError):
        sd_tiny.sample_latent(ctx, unc, 7.5, 0, init_latent=lat)  # n_steps = 0


def test_load_weights_dir_matches_set_weight(sd_tiny, synth, tiny_dims, tmp_path):
    """the npy-dump tree (reference format, python/save.py) read by the C++ loader
    (sdmi_load_weights_dir <- load_stable_diffusion, stablediffusion/load.rs:16-33) gives bit-identical
    results to feeding the same tensors through sdmi_set_weight."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, SdmiError, weights as wio
    d = tiny_dims
    specs = sd_tiny.weight_specs()
    shapes = dict(specs)

    def get(name, shape):
        parent, leaf = name.rsplit("/", 1)
        wshape = shapes.get(parent + "/weight")
        if leaf == "weight":
            if len(shape) == 4:
                return synth.get(name, shape, "w", shape[1] * shape[2] * shape[3])
            if len(shape) == 2:
                return synth.get(name, shape, "w", shape[0])
            return synth.get(name, shape, "gamma")
        if wshape is not None and len(wshape) == 4:
            return synth.get(name, shape, "b", wshape[1] * wshape[2] * wshape[3])
        if wshape is not None and len(wshape) == 2:
            return synth.get(name, shape, "b", wshape[0])
        return synth.get(name, shape, "beta")

    wio.write_dump_tree(tmp_path / "params", specs, get, syn.alphas_cumprod(), n_head=d.n_head)
    sd2 = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
    try:
        sd2.load_weights_dir(tmp_path / "params")
        lat, ctx, unc = _inputs(d, 1, 7, 2)
        a = sd_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
        b = sd2.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
        assert np.array_equal(a, b)
        # a missing file is a loud IO error, not a silent default
        (tmp_path / "params/unet/conv_out/bias.npy").unlink()
        sd3 = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
        with pytest.raises(SdmiError):
            sd3.load_weights_dir(tmp_path / "params")
        sd3.close()
    finally:
        sd2.close()


# ---- round 2: stream ordering of the *_dev entry points, batched weight loading, metadata, error clean-up -------------
def _dev_sample(sd, d, lat_t, ctx_t, unc_t, steps=2):
    out = torch.empty((lat_t.shape[0], 4, d.latent_h, d.latent_w), dtype=torch.float32, device="cuda")
    sd.sample_latent_dev(ctx_t.data_ptr(), lat_t.shape[0], ctx_t.shape[1], unc_t.data_ptr(), unc_t.shape[0], 7.5, steps,
                         lat_t.data_ptr(), out.data_ptr())
    return out


@pytest.mark.parametrize("mode", ["user_stream", "device_sync"])
def test_dev_entry_points_are_ordered_behind_the_producer(sd_tiny, tiny_dims, mode):
    """The engine works on a private non-blocking stream.  Inputs that are still being PRODUCED on another stream when
    sdmi_sample_latent_dev is called (here: a long chain of kernels on a torch side stream ending in the copy that fills
    the latent) must be waited for -- through sdmi_set_stream's event hand-shake, or the default device-wide sync."""
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    ref = sd_tiny.sample_latent(ctx, unc, 7.5, 2, init_latent=lat)
    side = torch.cuda.Stream()
    ctx_t = torch.from_numpy(ctx).cuda()
    unc_t = torch.from_numpy(unc).cuda()
    src = torch.from_numpy(lat).cuda()
    lat_t = torch.full_like(src, float("nan"))
    junk = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    try:
        if mode == "user_stream":
            sd_tiny.set_stream(side.cuda_stream)
        with torch.cuda.stream(side):
            for _ in range(40):                      # ~tens of ms of queued work in front of the real producer
                junk = junk @ junk * 1e-4
            lat_t.copy_(src, non_blocking=True)      # the producer of the input, last in the queue
            out = _dev_sample(sd_tiny, d, lat_t, ctx_t, unc_t)
            got = out.cpu().numpy()                  # consumer on the same stream (ordered behind the results)
    finally:
        sd_tiny.set_stream(0, enable=False)
    assert np.isfinite(got).all()
    assert np.array_equal(got, ref)


def test_packed_weights_match_set_weight(sd_tiny, synth, tiny_dims):
    """sdmi_load_weights_packed (one flat image, one staged upload) gives bit-identical results to per-tensor set_weight."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    ref = sd_tiny.sample_latent(ctx, unc, 7.5, 2, init_latent=lat)
    sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
    try:
        flat = sd.pack_weights(synth, groups=1)
        sd.load_weights_packed(flat, groups=1)
        got = sd.sample_latent(ctx, unc, 7.5, 2, init_latent=lat)
        img = sd.sample_image(ctx, unc, 7.5, 2, init_latent=lat)
        with pytest.raises(Exception):
            sd.load_weights_packed(flat[:-1], groups=1)
    finally:
        sd.close()
    assert np.array_equal(got, ref)
    assert np.array_equal(img, sd_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=lat))


def test_dump_metadata_is_honoured_or_rejected(sd_tiny, synth, tiny_dims, tmp_path):
    """The reference's loaders read eps / n_group / stride / padding next to the tensors (groupnorm/load.rs:15-19,
    load.rs:118-160).  A dump whose GroupNorm eps differs must change the result exactly like the oracle with that eps;
    a dump whose conv stride differs from what this engine hard-wires must be refused, not silently mis-run."""
    from stable_diffusion_burn_amd import ModelConfig, SdmiError, StableDiffusion, weights as W
    d = tiny_dims
    cfg = ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch)
    sd = StableDiffusion(cfg)
    try:
        specs = [(n, s) for n, s in sd.weight_specs() if not n.startswith("clip/") and not n.startswith("autoencoder/encoder/") and not n.startswith("autoencoder/quant_conv/")]
        shapes = dict(specs)
        W.write_dump_tree(tmp_path, specs, lambda n, s: syn.named_tensor(synth, n, s, shapes), syn.alphas_cumprod(), n_head=d.n_head)
        # (1) a different eps on the UNet's output norm
        np.save(tmp_path / "unet/norm_out/eps.npy", W.encode_scalar(0.5))
        sd.load_weights_dir(tmp_path)
        lat, ctx, _ = _inputs(d, 1, 7, 2)
        got = sd.unet.forward(lat, [500], ctx)
        base = sd_tiny.unet.forward(lat, [500], ctx)
        assert np.abs(got - base).max() > 1e-3          # eps = 0.5 on O(1)-variance activations is visible
        o64 = O.UNetOracle(synth, d, torch.float64)
        ref = o64.forward(torch.from_numpy(lat), 500, torch.from_numpy(ctx), norm_out_eps=0.5).numpy()
        assert np.abs(got - ref).max() < 1e-4
        # (2) a stride the engine does not implement for that layer
        np.save(tmp_path / "unet/norm_out/eps.npy", W.encode_scalar(1e-5))
        np.save(tmp_path / "unet/input_blocks/d1/stride.npy", W.encode_tensor(np.array([1, 1], np.float32)))
        with pytest.raises(SdmiError) as ei:
            sd.load_weights_dir(tmp_path)
        assert "stride" in str(ei.value)
    finally:
        sd.close()


def test_failed_call_returns_its_pool_blocks(sd_tiny, tiny_dims):
    """A call that throws half way (here: a context wider than the model's ctx_dim is caught up front, so use a batch
    above max_batch = unlimited -> force an unsupported attention shape instead) must not leak activation blocks:
    the same failing call repeated 50 times leaves the next good call's result and the pool unchanged."""
    d = tiny_dims
    lat, ctx, unc = _inputs(d, 1, 7, 2)
    ref = sd_tiny.sample_latent(ctx, unc, 7.5, 1, init_latent=lat)
    from stable_diffusion_burn_amd import SdmiError
    q = np.zeros((1, 8, 96), np.float32)   # head dim 96: not a fused instance; nk = 5 is not a multiple of 32 -> throws inside attention()
    k = np.zeros((1, 5, 96), np.float32)
    for _ in range(50):
        with pytest.raises(SdmiError):
            sd_tiny.qkv_attention(q, k, k, None, 1)
    got = sd_tiny.sample_latent(ctx, unc, 7.5, 1, init_latent=lat)
    assert np.array_equal(got, ref)


def test_load_weights_mpk_matches_set_weight(sd_tiny, synth, tiny_dims, tmp_path):
    """The `burn` model type (NamedMpkFileRecorder record, sample/main.rs:27-34) read by the C++ MessagePack walker and
    staged straight from the mapping gives bit-identical results to per-tensor set_weight.  (The record is written by
    tools/mpk_to_dump.write_record with the burn 0.14 layout of csrc/mpk_reader.hpp: UNPINNED against a real file.)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from tools import mpk_to_dump as M
    from stable_diffusion_burn_amd import ModelConfig, SdmiError, StableDiffusion
    d = tiny_dims
    sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
    try:
        specs = [(n, s) for n, s in sd.weight_specs() if not n.startswith("autoencoder/encoder/") and not n.startswith("autoencoder/quant_conv/")]
        shapes = dict(specs)
        tensors = {n: (syn.alphas_cumprod(s[0]) if n == "alphas_cumprod" else syn.named_tensor(synth, n, s, shapes)) for n, s in specs}
        M.write_record(tensors, tmp_path / "tiny.mpk")
        sd.load_weights_mpk(tmp_path / "tiny.mpk")
        lat, ctx, unc = _inputs(d, 1, 7, 2)
        got = sd.sample_latent(ctx, unc, 7.5, 2, init_latent=lat)
        assert np.array_equal(got, sd_tiny.sample_latent(ctx, unc, 7.5, 2, init_latent=lat))
        # a record for a different model width is refused with the tensor's name
        bad = dict(tensors)
        bad["unet/conv_out/bias"] = np.zeros(5, np.float32)
        M.write_record(bad, tmp_path / "bad.mpk")
        with pytest.raises(SdmiError) as ei:
            sd.load_weights_mpk(tmp_path / "bad.mpk")
        assert "unet/conv_out/bias" in str(ei.value)
    finally:
        sd.close()


def test_sharded_sample_image_through_the_c_abi(sd_tiny, synth, tiny_dims):
    """sdmi_create_multi + sdmi_sample_image_sharded on EVERY device this box has (1 on the builder's box; an 8-GPU box runs N = 8):
    one RCCL communicator, ONE broadcast of the packed prompt per call, every image bit-equal to the single-context path --
    with explicit x_T and with noise keyed by the global image index (seed + i).  n is not a multiple of the device count, so the
    contiguous ranges differ in size."""
    import torch as _t
    from stable_diffusion_burn_amd import ModelConfig, MultiStableDiffusion
    d = tiny_dims
    n_dev = _t.cuda.device_count()
    m = MultiStableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch), devices=tuple(range(n_dev)))
    try:
        m.load_weights(synth)
        n = 2 * n_dev + 1
        lat = np.stack([syn.initial_latent(i, d.latent_h, d.latent_w) for i in range(n)])
        ctx = syn.cond_context(0, 7, d.ctx_dim)
        unc = syn.uncond_context(2, d.ctx_dim)
        got = m.sample_image(ctx, unc, 7.5, 2, n, init_latents=lat)
        ref = sd_tiny.sample_image(np.repeat(ctx[None], n, axis=0), unc, 7.5, 2, init_latent=lat)
        for i in range(n):
            assert np.array_equal(got[i], ref[i]), f"image {i} of {n} on {n_dev} device(s)"
        assert m.broadcast_count() == 1
        got_seed = m.sample_image(ctx, unc, 7.5, 2, n, seed=11)
        ref_seed = sd_tiny.sample_image(np.repeat(ctx[None], n, axis=0), unc, 7.5, 2, seed=11)
        assert np.array_equal(got_seed, ref_seed)
        assert m.broadcast_count() == 2
    finally:
        m.close()
