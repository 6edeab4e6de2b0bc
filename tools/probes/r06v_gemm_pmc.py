"""round 6, call v: three bf16 GEMM launches (deep-K 3x3 at the 32x32 level, K = 320 Linear at the 64x64 level with N = 320 and N = 960, CFG batch 32) and the d = 40 self attention, a few
iterations each, for rocprofv3 --pmc passes: what do the waves of these kernels wait for?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("bench_cold", 0)
for shape, k in (((32, 640, 32, 32, 640), 3), ((32, 320, 64, 64, 320), 1), ((32, 320, 64, 64, 960), 1), ((32, 1280, 16, 16, 1280), 3)):
    ms = sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=3)
    print(shape, k, f"{ms * 1e3:.1f} us")
print("attention", sd.bench_attention(16, 4096, 4096, 320, 8, iters=2) * 1e3, "us")
