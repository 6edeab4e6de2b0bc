"""Start-phase spread of the persistent bf16 tile loop (k_gemm_bf16x.hip, variant bits 8..15): short-K Linear shapes at M = 131 072, time against the spread."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
SPREADS = (0, 6, 12, 18, 24, 32, 48, 0)
for N, tile in ((2560, 101), (2560, 100), (960, 100), (320, 100), (1280, 101)):
    for K in (320, 640, 1280):
        row = []
        for sp in SPREADS:
            sd.set_option("gemm_bf16x_variant", str(5 | (sp << 8)))
            ms = sd.bench_conv(32, K, 64, 64, N, k=1, stride=1, upsample2x=0, tile_cfg=tile, splitk=1, iters=8)
            row.append(f"{sp}: {ms * 1e3:6.1f}")
        print(f"N={N:5d} K={K:5d} tile {tile}: " + "  ".join(row), flush=True)
sd.close()
