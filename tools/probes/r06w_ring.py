"""round 6, call w: the four-stage ring (gemm_bf16x_variant bit 1) against the two-stage tiles, per shape, hot and cold: 256 x 256 (tile 101) and 256 x 128 (102) on the deep-K
and short-K shapes of the batch-16 model; also against the engine's own choice for the shape (tile -1: possibly 256 x 320 or the kernel-row form)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
SH = [((32, 1280, 16, 16, 1280), 3), ((32, 2560, 16, 16, 1280), 3), ((32, 640, 32, 32, 640), 3), ((32, 1280, 32, 32, 640), 3), ((32, 320, 64, 64, 320), 3), ((32, 640, 64, 64, 320), 3),
      ((1, 512, 128, 128, 512), 3), ((1, 256, 256, 256, 256), 3), ((1, 512, 256, 256, 256), 3), ((32, 1280, 16, 16, 1280), 1), ((32, 5120, 16, 16, 1280), 1), ((32, 640, 32, 32, 640), 1), ((32, 2560, 32, 32, 640), 1)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, k in SH:
        row = []
        sd.set_option("gemm_bf16x_variant", 1)
        auto = sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4)
        row.append(f"engine {auto * 1e3:7.1f}")
        for tile in (101, 102):
            for sp in (1, 2):
                r = []
                for v in (1, 3):
                    sd.set_option("gemm_bf16x_variant", v)
                    try:
                        r.append(sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=tile, splitk=sp, iters=4) * 1e3)
                    except Exception:  # noqa: BLE001
                        r.append(float("nan"))
                row.append(f"t{tile}x{sp}: {r[0]:7.1f} -> ring {r[1]:7.1f}")
        print(("cold " if cold else "hot  ") + f"{str(shape):30s} k{k}  " + "  ".join(row), flush=True)
