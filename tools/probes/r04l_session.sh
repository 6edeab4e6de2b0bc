#!/bin/bash
# round 4, call l: 128-row 4-wave bf16 tiles with two workgroups per CU (k_gemm_bf16s.hip, tiles 104 / 105) -- parity, then the Linear / 1x1 shapes of the batch-16 model against tiles 100 / 101 / 103
out=gpurun_out/r04l; mkdir -p $out
timeout 600 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x -k "large_tiles" > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log | cut -c1-300
timeout 900 python - > $out/bench.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
# (n, cin, h, w, cout, k): linears as 1x1 convs over [1, M] rows; M = 131072 / 32768 / 8192 / 2048 = the batch-16 CFG levels
CASES = [(1, 320, 1, 131072, 2560, 1), (1, 320, 1, 131072, 960, 1), (1, 320, 1, 131072, 320, 1), (1, 1280, 1, 131072, 320, 1),
         (1, 640, 1, 32768, 5120, 1), (1, 640, 1, 32768, 1920, 1), (1, 640, 1, 32768, 640, 1), (1, 2560, 1, 32768, 640, 1),
         (1, 1280, 1, 8192, 10240, 1), (1, 1280, 1, 8192, 3840, 1), (1, 1280, 1, 8192, 1280, 1), (1, 5120, 1, 8192, 1280, 1),
         (32, 320, 64, 64, 320, 3), (32, 1280, 16, 16, 1280, 3), (32, 1280, 8, 8, 1280, 3),
         (1, 320, 1, 65536, 2560, 1), (1, 320, 1, 65536, 320, 1), (1, 640, 1, 16384, 640, 1), (1, 1280, 1, 4096, 1280, 1)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for (n, cin, h, w, cout, k) in CASES:
        fl = 2.0 * n * h * w * cout * cin * k * k
        row = f"cold={cold} n={n} cin={cin} {h}x{w} cout={cout} k={k}:"
        for tile in (100, 101, 103, 104, 105):
            try:
                ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=tile, splitk=1, iters=8)
                row += f"  {tile}: {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF"
            except Exception as e:
                row += f"  {tile}: err"
        print(row, flush=True)
sd.close()
PY
echo "bench rc=$?"; grep -v amdgpu.ids $out/bench.txt | cut -c1-260
