"""round 6, call q: the MXFP8 16x16-level ResBlock convolution (M = 8192, N = 1280, K = 11520 / 23040 at CFG batch 32) under each tile / split-K, cold and hot -- the bf16
sweep (r06p) found 256x320 with split-K 2 19 % faster than the engine's choice for the same shape"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=2))
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape in ((32, 1280, 16, 16, 1280), (32, 2560, 16, 16, 1280), (32, 640, 32, 32, 640), (32, 1280, 8, 8, 1280)):
        row = []
        for tile, sp in ((-1, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 1), (2, 2), (0, 3), (0, 4)):
            try:
                ms = sd.bench_conv(*shape, k=3, stride=1, upsample2x=0, tile_cfg=tile, splitk=sp, iters=5)
                row.append(f"t{tile}x{sp}: {ms * 1e3:6.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"t{tile}x{sp}: err")
        print(("cold " if cold else "hot  ") + str(shape) + "  " + "  ".join(row), flush=True)
