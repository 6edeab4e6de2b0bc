#!/bin/bash
# round 4, call f: does the shape of an LDS-DMA piece (16 rows x 64 B vs 8 rows x 128 B) set the DMA rate?  timing only (ablate bit 3 gives wrong results)
out=gpurun_out/r04f; mkdir -p $out
timeout 600 python - > $out/ablate.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
CASES = [(32, 640, 64, 64, 320, 3), (32, 1280, 64, 64, 320, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    row = f"n={n} cin={cin} {h}x{w} cout={cout} k={k}:"
    for ab in (0, 8, 6, 14, 7):
        sd.set_option("gemm_ablate", ab)
        ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=104, splitk=1, iters=10)
        row += f"  ablate={ab}: {ms * 1e3:7.1f} us ({fl / ms / 1e9:6.0f})"
    sd.set_option("gemm_ablate", 0)
    print(row, flush=True)
sd.close()
PY
echo "rc=$?"; grep -v amdgpu.ids $out/ablate.txt | cut -c1-400
