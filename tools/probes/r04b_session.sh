#!/bin/bash
# round 4, call b: the ping-pong bf16 GEMM (k_gemm_bf16p.hip, bf16 tiles 104 / 105) -- parity, then per shape against the 8-wave tiles 100 / 101
out=gpurun_out/r04b; mkdir -p $out
timeout 600 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x -k "large_tiles" > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $out/pytest.log | cut -c1-300
timeout 600 python - > $out/bench.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
CASES = [(32, 320, 64, 64, 320, 3), (32, 640, 64, 64, 320, 3), (32, 640, 32, 32, 640, 3), (32, 1280, 32, 32, 640, 3), (32, 1280, 16, 16, 1280, 3), (32, 2560, 16, 16, 1280, 3),
         (32, 320, 64, 64, 320, 1), (32, 320, 64, 64, 2560, 1), (32, 1280, 64, 64, 320, 1), (16, 256, 256, 256, 256, 3), (16, 512, 64, 64, 512, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    row = f"n={n} cin={cin} {h}x{w} cout={cout} k={k}:"
    for tile in (100, 104, 101, 105):
        try:
            ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=tile, splitk=1, iters=10)
            row += f"  {tile}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s"
        except Exception as e:
            row += f"  {tile}: {e}"
    print(row, flush=True)
sd.close()
PY
echo "bench rc=$?"; grep -v amdgpu.ids $out/bench.txt | cut -c1-260
