#!/bin/bash
# round 4, call c: what bounds the bf16 k loop?  Ablations of k_gemm_bf16p.hip (tile 104): no DMA / no matrix instructions / no fragment reads (results are garbage, times are not)
out=gpurun_out/r04c; mkdir -p $out
timeout 600 python - > $out/ablate.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
CASES = [(32, 640, 64, 64, 320, 3), (32, 1280, 32, 32, 640, 3), (32, 320, 64, 64, 320, 1), (32, 1280, 64, 64, 320, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    row = f"n={n} cin={cin} {h}x{w} cout={cout} k={k}:"
    for ab in (0, 1, 2, 4, 3, 5, 6, 7):
        sd.set_option("gemm_ablate", ab)
        ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=104, splitk=1, iters=10)
        row += f"  ablate={ab}: {ms * 1e3:7.1f} us ({fl / ms / 1e9:6.0f})"
    sd.set_option("gemm_ablate", 0)
    print(row, flush=True)
sd.close()
PY
echo "rc=$?"; grep -v amdgpu.ids $out/ablate.txt | cut -c1-400
