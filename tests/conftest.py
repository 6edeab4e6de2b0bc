"""pytest configuration.

  -m "not gpu"  : oracle vs golden vectors, host logic, C-ABI symbol export (runs anywhere)
  -m gpu        : parity of the HIP path (through the C ABI) against the oracle; needs an MI355X
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s")


def gpu_available() -> bool:
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def tiny_dims():
    """Half-width model: same topology and head dims (40/80/160) as SD v1.4, 16x smaller maps."""
    from oracle.sd_oracle import Dims
    return Dims(model_channels=160, n_head=4, ctx_dim=64, latent_h=16, latent_w=16, vae_ch=32)


@pytest.fixture(scope="session")
def synth():
    from stable_diffusion_burn_amd.synthetic import SyntheticWeights
    return SyntheticWeights(cache=True)


@pytest.fixture(scope="session")
def sd_tiny(tiny_dims, synth):
    """HIP engine at the tiny dims with synthetic weights (GPU only)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    d = tiny_dims
    sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
    sd.load_weights(synth)
    yield sd
    sd.close()


@pytest.fixture(scope="session")
def sd_ops():
    """Engine used only for operator-level entry points (no weights needed)."""
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
    sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
    yield sd
    sd.close()
