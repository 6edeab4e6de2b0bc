"""Full-resolution float RGB of BASELINE.json configs[1] for the fp32 golden test (round 6: the float image was compared on a stride-4 grid only).

    python tests/golden/gen_golden_rgb_full.py        # ~1 min on 8 vCPU: ONE fp32 oracle decode of the committed final latent

  sd14_synth_cfg2_rgb_full.npz
       rgb32_q   [3,512,512] int16   round(rgb32 * 2^SHIFT): the fp32 oracle's decode (autoencoder/mod.rs:68-71 via oracle/sd_oracle.py decode_float) of
                                     sd14_synth_cfg2.npz latents32[-1], every pixel, fixed point -- quantisation error <= 2^-(SHIFT+1) (6.1e-5 at SHIFT = 13)
       shift     ()          int     SHIFT
The stride-4 samples of the same image in sd14_synth_cfg2.npz (rgb32_s4, exact fp32) must agree with this file to the quantisation step: checked here and in
tests/test_oracle_cpu.py.  Inputs: the seeded synthetic weights; nothing reads /root/reference.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle.sd_oracle import Dims, StableDiffusionOracle  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402

OUT = Path(__file__).resolve().parent
SHIFT = 13


def main():
    g = np.load(OUT / "sd14_synth_cfg2.npz")
    o32 = StableDiffusionOracle(syn.SyntheticWeights(cache=True), syn.alphas_cumprod(), Dims(), torch.float32)
    rgb = o32.decode_float(torch.from_numpy(g["latents32"][-1])[None])[0].numpy()
    assert np.abs(rgb).max() * (1 << SHIFT) < 32767, np.abs(rgb).max()
    q = np.rint(rgb.astype(np.float64) * (1 << SHIFT)).astype(np.int16)
    back = q.astype(np.float64) / (1 << SHIFT)
    print("quantisation error", np.abs(back - rgb).max(), "stride-4 agreement", np.abs(back[:, ::4, ::4] - g["rgb32_s4"]).max(), "absmax", np.abs(rgb).max())
    assert np.abs(back[:, ::4, ::4] - g["rgb32_s4"]).max() <= 2.0 ** -(SHIFT + 1) + 1e-7
    np.savez_compressed(OUT / "sd14_synth_cfg2_rgb_full.npz", rgb32_q=q, shift=np.array(SHIFT))


if __name__ == "__main__":
    main()
