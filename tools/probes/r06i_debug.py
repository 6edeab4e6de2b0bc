"""round 6, call i: which large-tile form disagrees when bias / time-embedding row enter through the accumulators (resid_acc bit 1)?  One batch-32 bf16 UNet forward
under conv3_reuse x resid_acc x gemm_bf16x_variant, compared with each other and with the resid_acc = 0 result."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn

sd = StableDiffusion(ModelConfig(precision=1))
sd.load_weights(syn.SyntheticWeights(), clip=False, vae_encoder=False)
n = 32
lat = np.stack([syn.initial_latent(i % 4) for i in range(n)])
ctx = np.stack([syn.cond_context(i % 4) for i in range(n)])
res = {}
for ra in (0, 1, 2, 3):
    for c3 in (1, 0):
        for pv in (1, 0):
            sd.set_option("resid_acc", ra); sd.set_option("conv3_reuse", c3); sd.set_option("gemm_bf16x_variant", pv)
            a = sd.unet.forward(lat, [500], ctx).astype(np.float64)
            b = sd.unet.forward(lat, [500], ctx).astype(np.float64)
            res[(ra, c3, pv)] = a
            base = res[(0, 1, 1)]
            print(f"resid_acc={ra} conv3_reuse={c3} persistent={pv}: repeat max|d| = {np.abs(a - b).max():.3e}; vs resid_acc=0 rel-RMS = "
                  f"{np.sqrt(np.mean((a - base) ** 2) / np.mean(base ** 2)):.3e} max|d| = {np.abs(a - base).max():.3e}", flush=True)
