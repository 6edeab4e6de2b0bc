"""CLIP text encoder + prompt->context on the GPU against the oracle and against the
reference's own Python model (tests/golden/refpy_clip.npz), through the C ABI.

Tolerance: fp32 kernels, |gpu - f64| <= 2e-5 * max(1, |ref|_inf) like the other operator tests (12 layers deep:
measured ~3e-6)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import clip_oracle as CO
from oracle.tokenizer_oracle import TokenizerOracle
from stable_diffusion_burn_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
MINI = GOLD / "mini_merges.txt"
MINI_VOCAB = 512 + 264 + 2


def _close(got, ref, what, rel=2e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    err = np.abs(got - ref).max()
    bound = rel * max(1.0, np.abs(ref).max())
    assert err <= bound, f"{what}: max|d| = {err:.3e} > {bound:.3e}"
    return err


@pytest.fixture(scope="module")
def sd_clip_tiny():
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(160, 4, 64, 16, 16, 32, clip_layers=2, clip_heads=1, clip_vocab=MINI_VOCAB, clip_ctx=16))
    sd.load_weights(syn.SyntheticWeights())
    yield sd
    sd.close()


TINY = CO.ClipDims(n_vocab=MINI_VOCAB, n_state=64, n_head=1, n_ctx=16, n_layer=2)


@pytest.mark.parametrize("n,T", [(1, 1), (1, 2), (2, 5), (3, 16)])
def test_clip_forward_tiny(sd_clip_tiny, n, T):
    g = np.random.default_rng(n * 100 + T)
    tokens = g.integers(0, MINI_VOCAB, (n, T)).astype(np.int32)
    got = sd_clip_tiny.clip.forward(tokens)
    ref = CO.CLIPOracle(syn.SyntheticWeights(), TINY, torch.float64).forward(tokens).numpy()
    _close(got, ref, f"clip tiny n={n} T={T}")


def test_context_matches_oracle(sd_clip_tiny):
    """StableDiffusion::context / unconditional_context (stablediffusion/mod.rs:194-210): tokenizer + CLIP, unpadded."""
    from stable_diffusion_burn_amd import SimpleTokenizer
    tok, ora = SimpleTokenizer(MINI), TokenizerOracle(MINI)
    clip64 = CO.CLIPOracle(syn.SyntheticWeights(), TINY, torch.float64)
    for text in ["a photo of a cat", ""]:
        ids = ora.context_tokens(text)
        got = sd_clip_tiny.context(tok, text)
        assert got.shape == (1, len(ids), 64)
        _close(got, clip64.forward(np.array([ids])).numpy(), f"context {text!r}")
    unc = sd_clip_tiny.unconditional_context(tok)
    assert unc.shape == (2, 64)
    np.testing.assert_array_equal(unc, sd_clip_tiny.context(tok, "")[0])


def test_context_feeds_sampling(sd_clip_tiny):
    """prompt -> context -> sample_image runs end to end on the device path (the reference's main.rs:100-109 sequence)."""
    from stable_diffusion_burn_amd import SimpleTokenizer
    tok = SimpleTokenizer(MINI)
    ctx = sd_clip_tiny.context(tok, "a painting of the sea at night")
    unc = sd_clip_tiny.unconditional_context(tok)
    img = sd_clip_tiny.sample_image(ctx, unc, 7.5, 2, init_latent=syn.initial_latent(0, 16, 16)[None])
    assert img.shape == (1, 128, 128, 3) and img.dtype == np.uint8 and img.std() > 1


def test_clip_errors(sd_clip_tiny):
    from stable_diffusion_burn_amd import ModelConfig, SdmiError, SimpleTokenizer, StableDiffusion
    with pytest.raises(SdmiError):
        sd_clip_tiny.clip.forward(np.zeros((1, 17), np.int32))              # longer than n_ctx
    with pytest.raises(SdmiError):
        sd_clip_tiny.clip.forward(np.full((1, 3), MINI_VOCAB, np.int32))    # id outside the table
    with pytest.raises(SdmiError):
        sd_clip_tiny.context(SimpleTokenizer(MINI), "a " * 40)               # 82 tokens > n_ctx = 16
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, clip_layers=2, clip_heads=1, clip_vocab=MINI_VOCAB, clip_ctx=16))
    try:
        sd.load_weights(syn.SyntheticWeights(), clip=False)                  # hot path only: a complete context
        with pytest.raises(SdmiError):
            sd.clip.forward(np.zeros((1, 2), np.int32))                      # CLIP group not loaded
        sd.set_weight("clip/layer_norm/weight", np.ones(64, np.float32))
        from stable_diffusion_burn_amd._capi import check
        with pytest.raises(SdmiError):
            check(sd._lib.sdmi_finalize_weights(sd._ctx))                    # partially set group
    finally:
        sd.close()


def test_clip_is_fp32_in_bf16_contexts(sd_clip_tiny):
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1, clip_layers=2, clip_heads=1, clip_vocab=MINI_VOCAB, clip_ctx=16))
    try:
        sd.load_weights(syn.SyntheticWeights())
        tokens = np.array([[MINI_VOCAB - 2, 5, 300, 77, MINI_VOCAB - 1]], np.int32)
        np.testing.assert_array_equal(sd.clip.forward(tokens), sd_clip_tiny.clip.forward(tokens))
    finally:
        sd.close()


def test_clip_full_size_vs_reference_python():
    """SD v1.4 CLIP (49408 x 768, 12 heads, 12 layers) against python/dump.py's CLIPTextTransformer outputs."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    g = np.load(GOLD / "refpy_clip.npz")
    sd = StableDiffusion(ModelConfig(64, 1, 768, 8, 8, 64, clip_layers=12))
    try:
        sd.load_weights(syn.SyntheticWeights())
        for key in ("probe", "t2", "t17", "t77"):
            got = sd.clip.forward(g[f"{key}_tokens"][None])[0]
            ref = g[f"{key}_out"]
            err = _close(got[::8] if key == "t77" else got, ref, f"clip full {key}")
            print(f"{key}: max|gpu - reference python| = {err:.2e} (|ref|max {np.abs(ref).max():.2f})")
        # batch of two different lengths padded by the caller is NOT the reference's API; batch = same length
        toks = np.stack([g["t17_tokens"], g["t17_tokens"][::-1]])
        both = sd.clip.forward(toks)
        np.testing.assert_allclose(both[0], sd.clip.forward(toks[:1])[0], rtol=0, atol=1e-6)
    finally:
        sd.close()


def _decode_png(path):
    """minimal PNG reader for 8-bit RGB, filter 0 (what sdmi_write_png emits); checks every CRC"""
    import struct
    import zlib
    data = Path(path).read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            w, h, depth, colour, comp, filt, inter = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, comp, filt, inter) == (8, 2, 0, 0, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_sample_cli_twin(sd_clip_tiny, tmp_path):
    """sdmi_sample (C++ twin of src/bin/sample/main.rs, same argv) on a dump tree == the same calls through Python."""
    import os
    import subprocess
    from stable_diffusion_burn_amd import SimpleTokenizer, build, weights as wio
    specs = sd_clip_tiny.weight_specs()
    shapes = dict(specs)
    W = syn.SyntheticWeights()
    wio.write_dump_tree(tmp_path / "params", specs, lambda name, shape: syn.named_tensor(W, name, shape, shapes),
                        syn.alphas_cumprod(), n_head=4, clip_heads=1)
    env = dict(os.environ, SDMI_BPE_VOCAB=str(MINI), SDMI_SEED="3",
               SDMI_CONFIG=f"model_channels=160,n_head=4,ctx_dim=64,latent_h=16,latent_w=16,vae_ch=32,clip_layers=2,clip_heads=1,clip_vocab={MINI_VOCAB},clip_ctx=16")
    out = tmp_path / "img"
    r = subprocess.run([str(build.CLI), "dump", str(tmp_path / "params"), "7.5", "2", "a photo of a cat", str(out), "hip:0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["Loading tokenizer...", "Loading model...", "Sampling image..."]   # main.rs:85-103
    got = _decode_png(str(out) + "0.png")
    tok = SimpleTokenizer(MINI)
    ref = sd_clip_tiny.sample_image(sd_clip_tiny.context(tok, "a photo of a cat"), sd_clip_tiny.unconditional_context(tok), 7.5, 2, seed=3)[0]
    np.testing.assert_array_equal(got, ref)
    # reference error conventions: bad arguments -> message on stderr, exit code 1 (main.rs:38-52,89-97)
    bad = subprocess.run([str(build.CLI), "dump", str(tmp_path / "nope"), "7.5", "2", "x", str(out), "hip:0"], env=env, capture_output=True, text=True)
    assert bad.returncode == 1 and "Error loading model dump" in bad.stderr
