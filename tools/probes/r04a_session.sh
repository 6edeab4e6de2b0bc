#!/bin/bash
# RECORD ONLY (the code it drives was removed after this ran: profiles/README.md "Round 4").  First GPU call of round 4 (written at the end of round 3, whose GPU minutes were spent): the staged two-term fp16 form of the fp32 GEMM
# (csrc/k_split2h.hip, k_gemm3p.hip NPL = 2, tile_cfg 400 + x).  1. its parity tests; 2. per shape, the plane tile the
# tuned table picks (300 + x, six bf16 products) against the same tile on two fp16 planes (400 + x, three products), operands hot and cold.
out=gpurun_out/r04a; mkdir -p $out
SDMI_STAGED=1 timeout 400 python -m pytest tests/test_f16s_staged_gpu.py -q -p no:cacheprovider -x -s > $out/pytest_staged.log 2>&1; echo "staged tests rc=$?"; tail -6 $out/pytest_staged.log | cut -c1-240
timeout 300 python - > $out/bench_300_vs_400.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
CASES = [((2, 320, 64, 64, 320, 3), (0, 4)), ((2, 640, 64, 64, 320, 3), (0, 4)), ((2, 640, 32, 32, 640, 3), (0, 8)), ((2, 1280, 16, 16, 1280, 3), (0, 16)),
         ((2, 1280, 8, 8, 1280, 3), (3, 32)), ((2, 320, 64, 64, 960, 1), (1, 1)), ((2, 320, 64, 64, 320, 1), (8, 1)), ((1, 256, 256, 256, 256, 3), (1, 1)), ((1, 512, 128, 128, 512, 3), (1, 1))]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for (n, cin, h, w, cout, k), (t, sp) in CASES:
        fl = 2.0 * n * h * w * cout * cin * k * k
        ms6 = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=300 + t, splitk=sp, iters=10)
        ms3 = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=400 + t, splitk=sp, iters=10) if not cold else float("nan")
        print(f"n={n} cin={cin} {h}x{w} cout={cout} k={k} tile x={t} splitk={sp} cold={cold}: six bf16 products {ms6 * 1e3:7.1f} us {fl / ms6 / 1e9:6.1f} TF/s | two fp16 terms {ms3 * 1e3:7.1f} us {fl / ms3 / 1e9:6.1f} TF/s", flush=True)
sd.close()
PY
echo "bench rc=$?"; cat $out/bench_300_vs_400.txt | grep -v amdgpu.ids | cut -c1-200
# 3. where the new kernel spends its time (gemm_probe: tiles 300 / 400 on the dominant batch-1 shape)
timeout 120 python - 2> $out/gemm_phase_probe_300_vs_400.txt <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
sd.set_option("gemm_probe", 1)
for tile in (300, 400, 303, 403):
    sd.bench_conv(2, 320, 64, 64, 320, k=3, tile_cfg=tile, splitk=4 if tile % 100 == 0 else 2, iters=5)
sd.close()
PY
grep gemm_probe $out/gemm_phase_probe_300_vs_400.txt | cut -c1-600
