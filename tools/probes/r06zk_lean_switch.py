"""3x3 kernel-row convolutions, lean epilogue on / off (gemm_bf16x_variant bit 3), hot and cold."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=prec))
SH = [((32, 320, 64, 64, 320), 3), ((32, 640, 32, 32, 640), 3), ((32, 1280, 32, 32, 640), 3), ((32, 1280, 16, 16, 1280), 3), ((32, 2560, 16, 16, 1280), 3), ((1, 128, 512, 512, 128), 3),
      ((32, 320, 64, 64, 320), 1), ((32, 640, 32, 32, 1920), 1)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, k in SH:
        r = []
        for v in (5, 13, 5, 13):
            sd.set_option("gemm_bf16x_variant", v)
            r.append(sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4) * 1e3)
        print(("cold " if cold else "hot  ") + f"{str(shape):30s} k{k}  lean {r[0]:7.1f} {r[2]:7.1f}   general {r[1]:7.1f} {r[3]:7.1f}", flush=True)
sd.close()
