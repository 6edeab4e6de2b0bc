"""Golden fixture for the UNPADDED-CONTEXT shape of the reference at the FULL model size: the reference's `context()`
does not pad the prompt to 77 tokens (stablediffusion/mod.rs:198-210), so a real call has Tc = tokens + 2 and Tu = 2 -- the conditional and
the unconditional halves of the CFG batch attend over DIFFERENT key counts.  The other full-size fixtures use T = Tu = 77.

    python tests/golden/gen_golden_q2.py          # ~12 min on 8 vCPU

  sd14_synth_q2.npz   B = 1, 20 DDIM steps, CFG 7.5, Tc = 77, Tu = 2 (the longest prompt against the empty one)
       latent32 [4,64,64] f32   final latent, fp32 oracle        latent64 [4,64,64] f64   final latent, fp64 oracle
       latents64_s [4,4,64,64]  fp64 latent after steps 1, 5, 10, 15
Inputs / weights: stable_diffusion_burn_amd/synthetic.py seeds (uncond_context(2): the first two rows of the seed-2 stream).  PARITY
UNPINNED in the sense of oracle/sd_oracle.py's header: this pins the GPU path to the oracle's restatement of the Rust lines.
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle.sd_oracle import Dims, StableDiffusionOracle  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    torch.set_num_threads(min(32, torch.get_num_threads()))
    d = Dims()
    w = syn.SyntheticWeights(cache=True)
    a = syn.alphas_cumprod()
    x = torch.from_numpy(syn.initial_latent(0))[None]
    ctx = torch.from_numpy(syn.cond_context(0, 77))[None]
    unc = torch.from_numpy(syn.uncond_context(2))
    t0 = time.time()
    o32 = StableDiffusionOracle(w, a, d, torch.float32)
    l32 = o32.sample_latent(ctx, unc, 7.5, 20, x)
    print(f"fp32 loop done ({time.time() - t0:.0f} s)", flush=True)
    o64 = StableDiffusionOracle(w, a, d, torch.float64)
    s64 = []
    l64 = o64.sample_latent(ctx, unc, 7.5, 20, x, per_step=s64)
    print(f"fp64 loop done ({time.time() - t0:.0f} s); |f32 - f64| = {float((l32.double() - l64).abs().max()):.2e}, absmax {float(l64.abs().max()):.1f}", flush=True)
    np.savez_compressed(OUT / "sd14_synth_q2.npz", latent32=l32.numpy()[0].astype(np.float32), latent64=l64.numpy()[0],
                        latents64_s=np.stack([s64[i].numpy()[0] for i in (0, 4, 9, 14)]))


if __name__ == "__main__":
    main()
