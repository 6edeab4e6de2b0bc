"""Where a plane-GEMM launch spends its time, per workgroup (option gemm_probe: every workgroup of conv_gemm3p_kernel stamps kernel entry,
first k tile landed, k loop done, epilogue stores acknowledged with the 100 MHz s_memrealtime clock; Engine::bench_conv prints the summary
on stderr).  Shapes = the batch-1 UNet's heaviest launch classes, with the tile / split-K the tuned table picks and its neighbours, operands
cache-hot and HBM-cold (bench_cold).

    python tools/probes/gemm_phase_probe.py 2> profiles/r03n_gemm_phase_probe.txt
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion          # noqa: E402

CASES = [
    # (n, cin, h, w, cout, k), [(tile, splitk), ...]   -- the probe instantiations exist for tiles 300 (256x160), 303 (128x160), 304 (128x128, 3 stages)
    ((2, 320, 64, 64, 320, 3), [(300, 4), (300, 2), (303, 2), (303, 1), (304, 2)]),
    ((2, 640, 64, 64, 320, 3), [(300, 4), (303, 2)]),
    ((2, 640, 32, 32, 640, 3), [(300, 8), (303, 4), (304, 4)]),
    ((2, 1280, 16, 16, 1280, 3), [(300, 16), (303, 8)]),
    ((2, 1280, 8, 8, 1280, 3), [(303, 32), (304, 32)]),
    ((2, 320, 64, 64, 320, 1), [(304, 1)]),
    ((2, 320, 64, 64, 960, 1), [(300, 1)]),
    ((1, 256, 256, 256, 256, 3), [(300, 1)]),
]

sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
sd.set_option("gemm_probe", 1)
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, tiles in CASES:
        n, cin, h, w, cout, k = shape
        for tile, sp in tiles:
            ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=tile, splitk=sp, iters=10)
            fl = 2.0 * n * h * w * cout * cin * k * k
            print(f"timed  n={n} cin={cin} {h}x{w} cout={cout} k={k} tile={tile} splitk={sp} cold={cold}: {ms * 1e3:.1f} us per conv (incl. reduce) "
                  f"{fl / ms / 1e9:.1f} TFLOP/s", file=sys.stderr, flush=True)
sd.close()
