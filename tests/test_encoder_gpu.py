"""VAE encoder on the GPU (Autoencoder::encode_image / forward, autoencoder/mod.rs:56-66)
against the oracle and the reference's Python model, through the C ABI."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import sd_oracle as O
from stable_diffusion_burn_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def _close(got, ref, what, rel=2e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    err = np.abs(got - ref).max()
    bound = rel * max(1.0, np.abs(ref).max())
    assert err <= bound, f"{what}: max|d| = {err:.3e} > {bound:.3e}"
    return err


def test_encode_image_tiny(sd_tiny, synth, tiny_dims):
    d = tiny_dims
    img = np.random.default_rng(5).uniform(-1, 1, (2, 3, 8 * d.latent_h, 8 * d.latent_w)).astype(np.float32)
    got = sd_tiny.autoencoder.encode_image(img)
    ref = O.EncoderOracle(synth, d, torch.float64).encode_image(torch.from_numpy(img)).numpy()
    err = _close(got, ref, "encode_image tiny", 5e-5)
    print(f"encode_image: max|gpu - f64| = {err:.2e} (|ref|max {np.abs(ref).max():.2f})")
    # per-sample independence
    one = sd_tiny.autoencoder.encode_image(img[1:2])
    assert np.abs(one - got[1:2]).max() <= 1e-5


def test_autoencoder_forward_tiny(sd_tiny, synth, tiny_dims):
    """Autoencoder::forward (:56-58) = decode_latent(encode_image(x))."""
    d = tiny_dims
    img = np.random.default_rng(6).uniform(-1, 1, (1, 3, 8 * d.latent_h, 8 * d.latent_w)).astype(np.float32)
    got = sd_tiny.autoencoder.forward(img)
    o = O.EncoderOracle(synth, d, torch.float64)
    ref = o.decode_latent(o.encode_image(torch.from_numpy(img))).numpy()
    _close(got, ref, "autoencoder forward tiny", 1e-4)


def test_encode_full_width_vs_reference_python():
    """full channel widths (128..512) on a 64 x 64 image against python/dump.py's encoder (refpy_encoder.npz)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    g = np.load(GOLD / "refpy_encoder.npz")
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 128))
    try:
        sd.load_weights(syn.SyntheticWeights())
        got = sd.autoencoder.encode_image(g["image"])
        err = _close(got, g["latent"], "encode_image full width", 5e-5)
        print(f"max|gpu - reference python| = {err:.2e} (|ref|max {np.abs(g['latent']).max():.2f})")
    finally:
        sd.close()


def test_encoder_group_is_optional(tiny_dims, synth):
    from stable_diffusion_burn_amd import ModelConfig, SdmiError, StableDiffusion
    d = tiny_dims
    sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
    try:
        sd.load_weights(synth, vae_encoder=False)
        img = np.zeros((1, 3, 8 * d.latent_h, 8 * d.latent_w), np.float32)
        with pytest.raises(SdmiError):
            sd.autoencoder.encode_image(img)
        z = sd.autoencoder.decode_latent(np.zeros((1, 4, d.latent_h, d.latent_w), np.float32))   # the hot path is complete
        assert np.isfinite(z).all()
    finally:
        sd.close()


def test_encode_bf16(tiny_dims):
    """precision = 1: relative RMS vs the fp64 oracle (bars as in test_bf16_gpu.py)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
    try:
        sd.load_weights(syn.SyntheticWeights())
        img = np.random.default_rng(7).uniform(-1, 1, (1, 3, 64, 64)).astype(np.float32)
        got = sd.autoencoder.encode_image(img)
        ref = O.EncoderOracle(syn.SyntheticWeights(), O.Dims(64, 1, 64, 8, 8, 64), torch.float64).encode_image(torch.from_numpy(img)).numpy()
        r = float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
        print(f"bf16 encode_image rel-RMS = {r:.3e}")
        assert np.isfinite(got).all() and r < 3e-2
    finally:
        sd.close()
