"""In-launch split-K combine (option splitk_coop, csk_combine in k_gemm_epi.hpp) against the separate reduce launch: the same convolutions and a
UNet forward of the half-width model with both, forced tiles / slice counts; repeated runs must be bit-identical (the sum runs in slice order
whatever the arrival order).  Prints the largest deviation between the two forms (they add the slices in different orders: <= a few fp32 ulps of
the accumulated magnitude) and the launches per forward."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion          # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn                     # noqa: E402

sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
rng = np.random.default_rng(5)
bad = 0
CASES = [  # n, cin, h, w, cout, k, tile, splitk
    (2, 320, 64, 64, 320, 3, 300, 4), (2, 320, 64, 64, 320, 3, 303, 2), (2, 640, 32, 32, 640, 3, 300, 8), (2, 1280, 16, 16, 1280, 3, 300, 16),
    (2, 1280, 8, 8, 1280, 3, 303, 32), (2, 1280, 8, 8, 1280, 3, 304, 32), (1, 320, 24, 40, 320, 3, 301, 3), (3, 64, 5, 7, 96, 3, 304, 2),
    (2, 640, 32, 32, 640, 1, 306, 2), (2, 320, 16, 16, 640, 3, 305, 5), (1, 256, 17, 19, 132, 3, 302, 4), (2, 320, 64, 64, 320, 3, 200, 4),
    (2, 320, 64, 64, 320, 3, 100, 4), (2, 960, 64, 64, 320, 3, 300, 4), (2, 2560, 8, 8, 1280, 3, 303, 32),
]
for n, cin, h, w, cout, k, tile, sp in CASES:
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    out = {}
    try:
        sd.set_option("gemm_tile", tile)
        sd.set_option("splitk", sp)
        for coop in (0, 1, 1):
            sd.set_option("splitk_coop", coop)
            t0 = time.time()
            y = sd.op_conv2d(x, wt, b)
            out.setdefault(coop, []).append(y)
    finally:
        sd.set_option("gemm_tile", "auto"); sd.set_option("splitk", 0); sd.set_option("splitk_coop", 1)
    ref = out[0][0]
    d = np.abs(out[1][0] - ref).max() / max(1.0, np.abs(ref).max())
    same = np.array_equal(out[1][0], out[1][1])
    ok = d < 2e-6 and same and np.isfinite(out[1][0]).all()
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} n={n} cin={cin} {h}x{w} cout={cout} k={k} tile={tile} splitk={sp}: max|coop - separate| / max|ref| = {d:.2e}, repeat identical: {same}", flush=True)
sd.close()

# the half-width model: residuals, time-embedding rows, planes outputs, every tile the tables pick
from oracle.sd_oracle import Dims                                            # noqa: E402  (shapes only)
d = Dims(model_channels=160, n_head=4, ctx_dim=64, latent_h=16, latent_w=16, vae_ch=32)
sd = StableDiffusion(ModelConfig(d.model_channels, d.n_head, d.ctx_dim, d.latent_h, d.latent_w, d.vae_ch))
sd.load_weights(syn.SyntheticWeights())
lat = np.stack([syn.initial_latent(i, 16, 16) for i in range(2)])
ctx = np.stack([syn.cond_context(i, 77, 64) for i in range(2)])
res = {}
for coop in (0, 1, 1):
    sd.set_option("splitk_coop", coop)
    y = sd.unet.forward(lat, [500], ctx)
    res.setdefault(coop, []).append((y, sd.last_call_stats()["kernels"]))
dd = np.abs(res[1][0][0] - res[0][0][0]).max() / np.abs(res[0][0][0]).max()
same = np.array_equal(res[1][0][0], res[1][1][0])
print(f"UNet forward (half width): max|coop - separate| / max = {dd:.2e}, repeat identical: {same}, kernels {res[0][0][1]} -> {res[1][0][1]}")
bad += not (dd < 1e-5 and same)
sd.close()
print("FAILED" if bad else "all ok")
sys.exit(1 if bad else 0)
