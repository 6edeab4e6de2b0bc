"""A few representative launches for a rocprofv3 --pmc pass (see profiles/README.md): conv / linear shapes per kernel family and the attention kernels."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

bf16 = "--bf16" in sys.argv
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1 if bf16 else 0))
if bf16:
    convs = [((32, 640, 32, 32, 640, 3, 1, 0), 100, 1), ((32, 320, 64, 64, 320, 3, 1, 0), 100, 1), ((32, 320, 64, 64, 320, 3, 1, 0), 0, 1),
             ((16, 256, 256, 256, 256, 3, 1, 0), 101, 1)]
    attn = [(16, 4096, 4096, 320, 8), (16, 1024, 1024, 640, 8)]
else:
    convs = [((1, 256, 256, 256, 256, 3, 1, 0), 101, 1), ((1, 256, 256, 256, 256, 3, 1, 0), 0, 1), ((2, 320, 64, 64, 320, 3, 1, 0), 9, 2),
             ((2, 320, 64, 64, 320, 3, 1, 0), 103, 4)]
    attn = [(2, 4096, 4096, 320, 8), (2, 1024, 1024, 640, 8)]
for s, cfg, sp in convs:
    ms = sd.bench_conv(*s, cfg, sp, 3)
    n, cin, h, w, cout, k = s[:6]
    print(s, cfg, sp, f"{ms * 1e3:.1f} us {2.0 * n * h * w * cout * cin * k * k / ms / 1e9:.0f} TF", flush=True)
for s in attn:
    ms = sd.bench_attention(*s, iters=3)
    n, nq, nk, c, hd = s
    print(s, f"{ms * 1e3:.1f} us {4.0 * n * hd * nq * nk * (c // hd) / ms / 1e9:.0f} TF", flush=True)
