"""VAE encoder oracle vs the reference's own Python model
(tests/golden/gen_encoder_from_reference_python.py -> refpy_encoder.npz)."""
from pathlib import Path

import numpy as np
import torch

from oracle import sd_oracle as O
from stable_diffusion_burn_amd import synthetic as syn

GOLD = Path(__file__).parent / "golden"


def test_encoder_oracle_matches_reference_python():
    g = np.load(GOLD / "refpy_encoder.npz")
    o = O.EncoderOracle(syn.SyntheticWeights(), O.Dims(), torch.float64)
    lat = o.encode_image(torch.from_numpy(g["image"])).numpy()
    err = np.abs(lat - g["latent"]).max()
    print(f"max|oracle - reference python| = {err:.3e}")
    assert lat.shape == (1, 4, 8, 8) and err < 1e-12


def test_autoencoder_probe_of_dump_py():
    """the commented probe of python/dump.py:613-619: autoencoder(zeros [1, 3, 10, 10]) -> [1, 3, 8, 8] (Autoencoder::forward)."""
    g = np.load(GOLD / "refpy_encoder.npz")
    o = O.EncoderOracle(syn.SyntheticWeights(), O.Dims(), torch.float64)
    out = o.decode_latent(o.encode_image(torch.zeros(1, 3, 10, 10))).numpy()
    assert out.shape == (1, 3, 8, 8)
    assert np.abs(out - g["probe_zeros_10x10"]).max() < 1e-12


def test_encoder_dump_names_are_the_reference_exporters():
    g = np.load(GOLD / "refpy_encoder.npz")
    ref_names = set(str(s) for s in g["dump_names"])
    asked = set()

    class Spy:
        def get(self, name, shape, kind, fan_in=0):
            asked.add(name)
            return np.zeros(shape, np.float32)

    O.EncoderOracle(Spy(), O.Dims(), torch.float32).encode_image(torch.zeros(1, 3, 16, 16))
    assert asked == ref_names, (sorted(asked - ref_names)[:5], sorted(ref_names - asked)[:5])


def test_padded_conv_is_the_reference_emulation():
    """PaddedConv2d (autoencoder/mod.rs:343-411): symmetric pad 2 + stride 2 + slice == asymmetric (0,1,0,1) padding."""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 12, 10, generator=gen, dtype=torch.float64)
    w = torch.randn(7, 5, 3, 3, generator=gen, dtype=torch.float64)
    b = torch.randn(7, generator=gen, dtype=torch.float64)
    full = F.conv2d(x, w, b, stride=2, padding=2)          # padding_actual = [2, 2]
    want = full[:, :, 1:1 + 6, 1:1 + 5]                     # skip = 1, desired = (0 + 1 + H - 3) / 2 + 1
    got = O.padded_conv2d(x, (w, b), 2, 0, 1, 0, 1)
    assert torch.equal(got, want)
