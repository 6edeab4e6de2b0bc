"""Launches for a rocprofv3 --pmc pass over the k-loop variants of the split (fp32) and bf16 large-tile GEMMs: every variant is its own
template instance, so the counter rows separate by kernel name.  Prints un-profiled-style timings too (under the profiler: slower clocks)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

bf16 = "--bf16" in sys.argv
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1 if bf16 else 0))
if bf16:
    shapes = [((16, 320, 64, 64, 320, 3, 1, 0), 100, 1), ((16, 640, 32, 32, 640, 3, 1, 0), 100, 1)]
    variants, key = (0, 3), "gemm_bf16x_variant"   # (round 2 also measured 1: the pipelined loop with hipcc's waits, since removed)
else:
    shapes = [((2, 320, 64, 64, 320, 3, 1, 0), 201, 4), ((2, 1280, 16, 16, 1280, 3, 1, 0), 204, 8), ((2, 640, 32, 32, 640, 3, 1, 0), 200, 8)]
    variants, key = (2, 74), "gemm3x_variant"        # (round 2 also measured 42: hoisted head + plane prefetch with hipcc's waits, since removed)
for v in variants:
    sd.set_option(key, v)
    for s, cfg, sp in shapes:
        ms = sd.bench_conv(*s, cfg, sp, 4)
        n, cin, h, w, cout, k = s[:6]
        print(s, cfg, sp, f"{key}={v}: {ms * 1e3:.1f} us {2.0 * n * h * w * cout * cin * k * k / ms / 1e9:.0f} TF", flush=True)
