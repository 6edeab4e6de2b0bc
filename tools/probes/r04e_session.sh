#!/bin/bash
# round 4, call e: ping-pong bf16 GEMM with the DMA issued in the matrix phase (tiles 106 / 107) -- parity on one test family, probe, per-shape bench vs 100 / 104
out=gpurun_out/r04e; mkdir -p $out
timeout 600 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x -k "large_tiles" > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log | cut -c1-300
timeout 600 python - > $out/bench.txt 2> $out/pp_probe.txt <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
sd.set_option("gemm_probe", 1)
for t in (104, 106):
    print(f"tile={t}", file=sys.stderr, flush=True)
    for (n, cin, h, w, cout, k) in [(32, 640, 64, 64, 320, 3), (32, 1280, 64, 64, 320, 1)]:
        sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=t, splitk=1, iters=3)
sd.set_option("gemm_probe", 0)
CASES = [(32, 320, 64, 64, 320, 3), (32, 640, 64, 64, 320, 3), (32, 640, 32, 32, 640, 3), (32, 1280, 32, 32, 640, 3), (32, 1280, 16, 16, 1280, 3), (32, 2560, 16, 16, 1280, 3),
         (32, 320, 64, 64, 320, 1), (32, 320, 64, 64, 2560, 1), (32, 1280, 64, 64, 320, 1), (16, 256, 256, 256, 256, 3), (16, 512, 64, 64, 512, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    row = f"n={n} cin={cin} {h}x{w} cout={cout} k={k}:"
    for tile in (100, 104, 106, 101, 107):
        ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=tile, splitk=1, iters=10)
        row += f"  {tile}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s"
    print(row, flush=True)
sd.close()
PY
echo "bench rc=$?"; grep "pp_probe\|tile=" $out/pp_probe.txt | cut -c1-400; grep -v amdgpu.ids $out/bench.txt | cut -c1-300
