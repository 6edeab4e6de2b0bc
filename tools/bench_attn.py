"""A/B the attention kernels on the UNet's self/cross-attention shapes (HIP events inside libsdmi)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

SHAPES = [(2, 4096, 4096, 320, 8), (2, 1024, 1024, 640, 8), (2, 256, 256, 1280, 8), (2, 64, 64, 1280, 8),
          (2, 4096, 77, 320, 8), (2, 1024, 77, 640, 8), (2, 256, 77, 1280, 8)]
sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
for s in SHAPES:
    n, nq, nk, c, h = s
    fl = 4.0 * n * h * nq * nk * (c // h)
    row = []
    for variant in (0, 1):
        sd.set_option("attn_variant", variant)
        ms = sd.bench_attention(*s, iters=10)
        row.append(f"v{variant}: {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF")
    print(f"{str(s):34s} " + " | ".join(row), flush=True)
