"""Phase shares of the ONE-TILE bf16 forms (instrumented build: tools/probes/r06zg_add_phase_probe.py) on the model's launches with at most 256 tiles, hot and cold."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("gemm_probe", 1)
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, tile in (((32, 1280, 16, 16, 1280), 103), ((32, 640, 32, 32, 640), 100), ((32, 1280, 16, 16, 3840), 101), ((32, 5120, 16, 16, 1280), 103), ((32, 2560, 32, 32, 640), 100), ((32, 320, 64, 64, 320), 100)):
        ms = sd.bench_conv(*shape, k=1, stride=1, upsample2x=0, tile_cfg=tile, splitk=1, iters=6)
        print(f"timed cold={cold} {shape} tile={tile}: {ms*1e3:.1f} us", file=sys.stderr, flush=True)
sd.close()
