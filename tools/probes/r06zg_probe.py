import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("gemm_probe", 1)
for N, tile in ((2560, 101), (2560, 100), (960, 100), (320, 100), (2560, 102)):
    for K in (64, 320, 1280):
        ms = sd.bench_conv(32, K, 64, 64, N, k=1, stride=1, upsample2x=0, tile_cfg=tile, splitk=1, iters=6)
        print(f"timed N={N} K={K} tile={tile}: {ms*1e3:.1f} us", file=sys.stderr, flush=True)
sd.close()
