"""round 6, call x: the large-tile bf16 GEMM with the DMA issue of waves 4 - 7 moved between the two k steps of a tile (gemm_bf16x_variant bit 2; 256 x 256 and 256 x 128 tiles), per shape"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
SH = [((32, 1280, 16, 16, 1280), 3), ((32, 640, 32, 32, 640), 3), ((32, 320, 64, 64, 320), 3), ((1, 256, 256, 256, 256), 3), ((1, 512, 256, 256, 256), 3),
      ((32, 1280, 16, 16, 1280), 1), ((32, 5120, 16, 16, 1280), 1), ((32, 640, 32, 32, 2560), 1), ((32, 320, 64, 64, 2560), 1)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, k in SH:
        row = []
        for tile in (101, 102):
            r = []
            for v in (1, 5, 1, 5):
                sd.set_option("gemm_bf16x_variant", v)
                r.append(sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=tile, splitk=1, iters=4) * 1e3)
            row.append(f"t{tile}: {r[0]:7.1f} {r[2]:7.1f} -> stagger {r[1]:7.1f} {r[3]:7.1f}")
        print(("cold " if cold else "hot  ") + f"{str(shape):30s} k{k}  " + "   ".join(row), flush=True)
