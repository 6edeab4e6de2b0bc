"""3x3 convolutions (kernel-row form, engine's tile choice) and the Linear / 1x1 shapes of the bf16 B = 16 model, hot and cold: one line per shape (library A/B: run under each build)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=prec))
SH = [((32, 320, 64, 64, 320), 3), ((32, 640, 64, 64, 320), 3), ((32, 960, 64, 64, 320), 3), ((32, 640, 32, 32, 640), 3), ((32, 1280, 32, 32, 640), 3), ((32, 1920, 32, 32, 640), 3),
      ((32, 1280, 16, 16, 1280), 3), ((32, 2560, 16, 16, 1280), 3), ((32, 1280, 8, 8, 1280), 3), ((16, 512, 64, 64, 512), 3), ((1, 512, 128, 128, 512), 3), ((1, 256, 256, 256, 256), 3), ((1, 128, 512, 512, 128), 3),
      ((32, 320, 64, 64, 320), 1), ((32, 320, 64, 64, 960), 1), ((32, 1280, 64, 64, 320), 1), ((32, 640, 32, 32, 640), 1), ((32, 640, 32, 32, 1920), 1), ((32, 2560, 32, 32, 640), 1),
      ((32, 1280, 16, 16, 1280), 1), ((32, 1280, 16, 16, 3840), 1), ((32, 5120, 16, 16, 1280), 1), ((32, 960, 64, 64, 320), 1), ((32, 1920, 32, 32, 640), 1)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, k in SH:
        r = [sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4) * 1e3 for _ in range(3)]
        print(("cold " if cold else "hot  ") + f"{str(shape):30s} k{k}  " + "  ".join(f"{x:7.1f}" for x in r), flush=True)
sd.close()
