"""Round 3, GPU call m: run the precision = 1 / 2 model-level cases on the GPU and SAVE their outputs (gpurun_out/r03m/*.npy), so that the
fp64-oracle side of the comparison (minutes of host time) is evaluated off the GPU box (tools/probes/r03m_compare.py) -- the box is charged
by wall time.  The tests (tests/test_fp8_gpu.py, tests/test_golden_gpu.py) do the same comparisons in one process; their bars are set from the
numbers this pair of scripts prints."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion          # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn                     # noqa: E402

out = ROOT / "gpurun_out" / "r03m"
out.mkdir(parents=True, exist_ok=True)
gold = np.load(ROOT / "tests" / "golden" / "sd14_synth_cfg5.npz")
t0 = time.time()


def inputs(n):
    lat = np.stack([syn.initial_latent(i) for i in range(n)])
    ctx = np.repeat(syn.cond_context(0)[None], n, axis=0)
    return lat, ctx, syn.uncond_context()


# ---- 8x8-latent, full-width UNet forward at precision 2 (tests/test_fp8_gpu.py::test_unet_forward_mxfp8)
sd = StableDiffusion(ModelConfig(320, 8, 768, 8, 8, 64, precision=2))
sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
sd.set_option("fp8_min_rows", 1)
lat = np.stack([syn.initial_latent(i, 8, 8) for i in range(2)])
ctx = np.stack([syn.cond_context(i, 77, 768) for i in range(2)])
for wide in (0, 1):
    sd.set_option("fp8_linear", wide)
    np.save(out / f"unet8_wide{wide}.npy", sd.unet.forward(lat, [999], ctx))
    print(f"unet8 wide={wide}: {sd.last_call_stats()['kernels']} kernels per forward", flush=True)
sd.set_option("fp8_convs", 0)
np.save(out / "unet8_bf16.npy", sd.unet.forward(lat, [999], ctx))
sd.close()
print(f"unet8 done {time.time() - t0:.0f} s", flush=True)

# ---- configs[4] at full size: batch 16, 20 steps, precision 2 (tests/test_golden_gpu.py::test_config5_mxfp8_batch16_20_steps)
sd = StableDiffusion(ModelConfig(precision=2))
sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
lat, ctx, unc = inputs(16)
lat[15] = lat[0]
for wide in (1, 0):
    sd.set_option("fp8_linear", wide)
    got = sd.sample_latent(ctx, unc, 7.5, 20, init_latent=lat)
    assert np.isfinite(got).all() and np.array_equal(got[0], got[15])
    np.save(out / f"cfg5_latent_wide{wide}.npy", got[:2])
    rgb = sd.autoencoder.decode_latent((gold["latent64"][:2] * (1.0 / 0.18215)).astype(np.float32))
    np.save(out / f"cfg5_rgb_s4_wide{wide}.npy", rgb[:, :, ::4, ::4])
    print(f"cfg5 wide={wide} done {time.time() - t0:.0f} s", flush=True)
sd.close()

# ---- configs[2] at full size: batch 16, 50 steps, bf16 (tests/test_golden_gpu.py::test_config3_bf16_batch16_50_steps)
sd = StableDiffusion(ModelConfig(precision=1))
sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
lat, ctx, unc = inputs(16)
got = sd.sample_latent(ctx, unc, 7.5, 50, init_latent=lat)
np.save(out / "cfg3_latent_bf16.npy", got[:2])
g3 = np.load(ROOT / "tests" / "golden" / "sd14_synth_cfg3.npz")
rgb = sd.autoencoder.decode_latent((g3["latent64"][:2] * (1.0 / 0.18215)).astype(np.float32))
np.save(out / "cfg3_rgb_s4_bf16.npy", rgb[:, :, ::4, ::4])
sd.close()
print(f"cfg3 bf16 done {time.time() - t0:.0f} s", flush=True)
