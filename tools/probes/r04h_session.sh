#!/bin/bash
# round 4, call h: 8-wave tiles 100 / 101 vs ping-pong 104 / 105 with HBM-cold operands (what a layer sees inside the model)
out=gpurun_out/r04h; mkdir -p $out
timeout 800 python - > $out/bench_cold.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("bench_cold", 1)
CASES = [(32, 320, 64, 64, 320, 3, 1), (32, 640, 32, 32, 640, 3, 1), (32, 1280, 16, 16, 1280, 3, 1), (32, 1280, 8, 8, 1280, 3, 4), (32, 2560, 16, 16, 1280, 3, 1),
         (32, 320, 64, 64, 2560, 1, 1), (32, 1280, 16, 16, 1280, 1, 1), (32, 640, 32, 32, 640, 1, 1)]
for (n, cin, h, w, cout, k, sp) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    row = f"cold n={n} cin={cin} {h}x{w} cout={cout} k={k} splitk={sp}:"
    for tile in (100, 104, 101, 105):
        ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=tile, splitk=sp, iters=8)
        row += f"  {tile}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s"
    print(row, flush=True)
sd.close()
PY
echo "rc=$?"; grep -v amdgpu.ids $out/bench_cold.txt | cut -c1-300
