#!/bin/bash
# round 4, call d: per-segment cycle sums inside the ping-pong bf16 k loop (option gemm_probe, tile 104)
out=gpurun_out/r04d; mkdir -p $out
timeout 600 python - 2> $out/pp_probe.txt <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("gemm_probe", 1)
for ab in (0, 1, 4):
    sd.set_option("gemm_ablate", ab)
    print(f"ablate={ab}", file=sys.stderr, flush=True)
    for (n, cin, h, w, cout, k) in [(32, 640, 64, 64, 320, 3), (32, 1280, 64, 64, 320, 1)]:
        sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=104, splitk=1, iters=3)
sd.close()
PY
echo "rc=$?"; grep "pp_probe\|ablate" $out/pp_probe.txt | cut -c1-400
