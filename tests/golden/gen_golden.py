"""Generate the committed golden fixtures for the FULL-SIZE model with the CPU oracle.

    python tests/golden/gen_golden.py            # ~10-15 min on 8 vCPU (fp32 + fp64 runs)

The GPU box has no /root/reference and should not spend minutes of host time per
test run, so the expensive oracle runs happen here once and the results are
committed as small .npz files:

  sd14_synth_cfg2.npz   BASELINE.json configs[1]: B=1, 20 DDIM steps, CFG 7.5, T=Tu=77
       latents32  [20,4,64,64] f32   latent after every step (fp32 oracle)
       latent64   [4,64,64]    f64   final latent (fp64 oracle)
       step_err   [20]         f64   max|f32-f64| after every step
       rgb_u8     [512,512,3]  u8    oracle image (from the fp32 latent)
       rgb32_s4 / rgb64_s4 [3,128,128]  decoded float RGB on a stride-4 grid
       rgb32_stats             per-channel mean/std/min/max of the full float image
  sd14_synth_cfg1.npz   configs[0]: B=1, 1 step, "CFG off" (scale 1.0): final latent + rgb
  sd14_synth_unet.npz   one UNet forward (t=999 and t=49) of x_T with the cond context

Inputs and weights are the seeded synthetic ones of stable_diffusion_burn_amd/synthetic.py
(BASELINE.md section 3); nothing here reads /root/reference.  PARITY UNPINNED: these
vectors pin the GPU path to THIS oracle, not to a run of the Rust reference (which
cannot be built here) -- see oracle/sd_oracle.py header and DESIGN.md.
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
    ctx = torch.from_numpy(syn.cond_context(0))[None]
    unc = torch.from_numpy(syn.uncond_context())
    o32 = StableDiffusionOracle(w, a, d, torch.float32)
    o64 = StableDiffusionOracle(w, a, d, torch.float64)

    t0 = time.time()
    u = {}
    for t in (999, 49):
        u[f"eps32_t{t}"] = o32.unet.forward(x, t, ctx).numpy()[0]
        u[f"eps64_t{t}"] = o64.unet.forward(x, t, ctx).numpy()[0]
        print(f"unet t={t}: |f32-f64| = {np.abs(u[f'eps32_t{t}'] - u[f'eps64_t{t}']).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(OUT / "sd14_synth_unet.npz", **u)

    # config 1: one step, scale 1.0
    l32 = o32.sample_latent(ctx, unc, 1.0, 1, x)
    l64 = o64.sample_latent(ctx, unc, 1.0, 1, x)
    img, f32img = o32.latent_to_image(l32)
    np.savez_compressed(OUT / "sd14_synth_cfg1.npz", latent32=l32.numpy()[0], latent64=l64.numpy()[0], rgb_u8=img[0])
    print(f"cfg1 done ({time.time() - t0:.0f} s)", flush=True)

    # config 2: 20 steps, CFG 7.5
    s32, s64 = [], []
    l32 = o32.sample_latent(ctx, unc, 7.5, 20, x, per_step=s32)
    print(f"cfg2 fp32 loop done ({time.time() - t0:.0f} s)", flush=True)
    l64 = o64.sample_latent(ctx, unc, 7.5, 20, x, per_step=s64)
    print(f"cfg2 fp64 loop done ({time.time() - t0:.0f} s)", flush=True)
    step_err = np.array([float((a_.double() - b_).abs().max()) for a_, b_ in zip(s32, s64)])
    rgb32 = o32.decode_float(l32)[0]
    rgb64 = o64.decode_float(l64)[0]
    img, _ = o32.latent_to_image(l32)
    r = rgb32.numpy()
    stats = np.stack([r.mean(axis=(1, 2)), r.std(axis=(1, 2)), r.min(axis=(1, 2)), r.max(axis=(1, 2))])
    np.savez_compressed(
        OUT / "sd14_synth_cfg2.npz",
        latents32=np.stack([s.numpy()[0] for s in s32]).astype(np.float32), latent64=l64.numpy()[0], step_err=step_err,
        rgb_u8=img[0], rgb32_s4=r[:, ::4, ::4].copy(), rgb64_s4=rgb64.numpy()[:, ::4, ::4].copy(), rgb32_stats=stats,
        rgb_f32_f64_maxdiff=np.array(float((rgb32.double() - rgb64).abs().max())))
    print(f"cfg2 done ({time.time() - t0:.0f} s): final |f32-f64| latent {step_err[-1]:.2e}, "
          f"rgb {float((rgb32.double() - rgb64).abs().max()):.2e}, latent absmax {float(l64.abs().max()):.1f}", flush=True)


if __name__ == "__main__":
    main()
