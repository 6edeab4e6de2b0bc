"""Does the bf16 tile choice still hold after the epilogue changes?  Linear / 1x1 shapes of the batch-16 UNet (n = 32): the engine's own choice against every large tile forced, HBM-cold."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32      # UNet rows: 2 x images per GPU
SH = [(32, 320, 64, 64, 320), (32, 320, 64, 64, 960), (32, 1280, 64, 64, 320), (32, 640, 64, 64, 320), (32, 960, 64, 64, 320),
      (32, 640, 32, 32, 640), (32, 640, 32, 32, 1920), (32, 2560, 32, 32, 640), (32, 1280, 32, 32, 640), (32, 1920, 32, 32, 640), (32, 960, 32, 32, 640),
      (32, 1280, 16, 16, 1280), (32, 1280, 16, 16, 3840), (32, 5120, 16, 16, 1280), (32, 2560, 16, 16, 1280), (32, 1920, 16, 16, 1280), (32, 640, 16, 16, 1280),
      (32, 1280, 8, 8, 1280), (32, 1280, 8, 8, 3840), (32, 5120, 8, 8, 1280), (16, 320, 64, 64, 320), (16, 320, 64, 64, 960)]
SH = [(NB * s[0] // 32,) + s[1:] for s in SH]
sd.set_option("bench_cold", 1)
for shape in SH:
    auto = min(sd.bench_conv(*shape, k=1, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4) for _ in range(2)) * 1e3
    row = []
    best = (auto, "auto")
    for tile in (100, 101, 102, 103):
        for sp in (1, 2):
            try:
                t = min(sd.bench_conv(*shape, k=1, stride=1, upsample2x=0, tile_cfg=tile, splitk=sp, iters=4) for _ in range(2)) * 1e3
            except Exception:
                continue
            row.append(f"{tile}/{sp}: {t:6.1f}")
            if t < best[0]: best = (t, f"{tile}/{sp}")
    flag = "  <-- " + best[1] + f" {100 * (auto / best[0] - 1):.1f} % faster" if best[0] < 0.97 * auto else ""
    print(f"{str(shape):28s} auto {auto:6.1f} | " + "  ".join(row) + flag, flush=True)
sd.close()
