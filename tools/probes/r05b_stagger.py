"""Round 5 probe: persistent bf16 GEMM with every other group of 8 workgroups started late (gemm_bf16x_variant bits 4..7 = units of s_sleep 64 ~ 2 us):
does de-phasing the CUs hide the epilogue's output burst behind the other half's k loops?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
for N, K in ((2560, 320), (960, 320), (320, 320), (320, 1280), (2560, 640)):
    row = []
    for stag in (None, 0, 2, 3, 4, 5, 6, 8, 10):
        sd.set_option("gemm_bf16x_variant", 0 if stag is None else 1 + 16 * stag)
        ms = sd.bench_conv(32, K, 64, 64, N, k=1, stride=1, upsample2x=0, tile_cfg=100, splitk=1, iters=8)
        row.append(f"{'v0' if stag is None else 's%d' % stag}: {ms * 1e3:6.1f}")
    print(f"N={N:5d} K={K:5d} tile 100 (us): " + "  ".join(row), flush=True)
sd.close()
