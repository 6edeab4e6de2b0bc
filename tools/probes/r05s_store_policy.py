"""Round 5 probe: cache policy of the bf16 epilogue's output stores (plain / nt / sc1 write-through) on the short-K Linear shapes, with and without the persistent tile loop."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
for rep in range(2):
    for N, K in ((2560, 320), (960, 320), (320, 320), (320, 1280), (2560, 640)):
        row = []
        for name, v in (("plain", 1), ("nt", 5), ("sc1", 9), ("plain-1tile", 0), ("nt-1tile", 4), ("sc1-1tile", 8)):
            sd.set_option("gemm_bf16x_variant", v)
            ms = sd.bench_conv(32, K, 64, 64, N, k=1, stride=1, upsample2x=0, tile_cfg=100, splitk=1, iters=8)
            row.append(f"{name}: {ms * 1e3:6.1f}")
        print(f"N={N:5d} K={K:5d} tile 100 (us): " + "  ".join(row), flush=True)
sd.close()
