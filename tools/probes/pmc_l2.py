"""L2 hit rate of the split GEMM under the legacy block -> tile map and under the XCD cut (option xcd_map): launches for
`rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum`.  Same kernel instance in both arms: the rows separate by dispatch order (all legacy launches first)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=0))
shapes = [((2, 320, 64, 64, 320, 3, 1, 0), 201, 4), ((2, 1280, 16, 16, 1280, 3, 1, 0), 204, 8), ((2, 640, 32, 32, 640, 3, 1, 0), 200, 8)]
for xm in (0, 1):
    sd.set_option("xcd_map", xm)
    for s, cfg, sp in shapes:
        ms = sd.bench_conv(*s, cfg, sp, 3)
        print(s, cfg, sp, f"xcd_map={xm}: {ms * 1e3:.1f} us", flush=True)
