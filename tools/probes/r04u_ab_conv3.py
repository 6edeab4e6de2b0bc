"""A/B of the kernel-row tiles (k_gemm_bf16t.hip, 104 / 105) against the tiles they replace (100 / 101) on the 3x3 / stride-1 shapes of the batch-8 / batch-16 bf16 model,
operands hot and HBM-cold."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
# (n, cin, h, w, cout): UNet 3x3 convolutions at CFG batch n = 2 B, and the decoder's 64x64 / 128x128 levels (n = 1)
SHAPES = []
for n in (32, 16):
    SHAPES += [(n, 320, 64, 64, 320), (n, 640, 64, 64, 320), (n, 960, 64, 64, 320), (n, 320, 32, 32, 640), (n, 640, 32, 32, 640), (n, 960, 32, 32, 640),
               (n, 1280, 32, 32, 640), (n, 1920, 32, 32, 640), (n, 640, 16, 16, 1280), (n, 1280, 16, 16, 1280), (n, 1920, 16, 16, 1280), (n, 2560, 16, 16, 1280)]
SHAPES += [(1, 512, 64, 64, 512), (1, 512, 128, 128, 512)]
for sh in SHAPES:
    fl = 2.0 * sh[0] * sh[2] * sh[3] * sh[4] * sh[1] * 9
    line = f"{str(sh):28s}"
    for cold in (0, 1):
        sd.set_option("bench_cold", cold)
        it = 6 if fl < 5e11 else 3
        auto = sd.bench_conv(*sh, k=3, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=it)
        r = {}
        for t in (100, 104, 101, 105):
            try:
                r[t] = sd.bench_conv(*sh, k=3, stride=1, upsample2x=0, tile_cfg=t, splitk=1, iters=it)
            except Exception as e:
                r[t] = float("nan")
        line += f" | {'cold' if cold else 'hot '}: auto {auto * 1e3:7.1f} us  100 {r[100] * 1e3:7.1f} -> 104 {r[104] * 1e3:7.1f} ({r[100] / r[104]:.3f}x, {fl / r[104] / 1e9:.0f} TF)  101 {r[101] * 1e3:7.1f} -> 105 {r[105] * 1e3:7.1f} ({r[101] / r[105]:.3f}x)"
    print(line, flush=True)
sd.close()
