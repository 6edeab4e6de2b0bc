"""Launches for a rocprofv3 --pmc pass over the split-operand kernels (k_gemm3x.hip, k_attn_split.hip): see profiles/README.md."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=0))
convs = [((2, 320, 64, 64, 320, 3, 1, 0), 200, 4), ((2, 640, 64, 64, 320, 3, 1, 0), 200, 4), ((1, 512, 128, 128, 512, 3, 1, 0), 203, 1),
         ((1, 256, 256, 256, 256, 3, 1, 0), 202, 1), ((2, 320, 64, 64, 2560, 1, 1, 0), 200, 1), ((2, 320, 64, 64, 320, 1, 1, 0), 205, 1),
         ((1, 256, 256, 256, 256, 3, 1, 0), 101, 1)]
for variant in (0, 1):
    sd.set_option("gemm3x_variant", variant)
    for s, cfg, sp in convs:
        if variant and cfg < 200:
            continue
        ms = sd.bench_conv(*s, cfg, sp, 3)
        n, cin, h, w, cout, k = s[:6]
        print(s, cfg, sp, f"variant {variant}: {ms * 1e3:.1f} us {2.0 * n * h * w * cout * cin * k * k / ms / 1e9:.0f} TF", flush=True)
for mode in (1, 2, 0):
    sd.set_option("attn_split", mode)
    for s in [(2, 4096, 4096, 320, 8), (2, 1024, 1024, 640, 8)]:
        ms = sd.bench_attention(*s, iters=3)
        n, nq, nk, c, hd = s
        print(s, f"attn_split {mode}: {ms * 1e3:.1f} us {4.0 * n * hd * nq * nk * (c // hd) / ms / 1e9:.0f} TF", flush=True)
