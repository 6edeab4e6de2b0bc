#!/bin/bash
# round 5, probe x: k_gemm3w.hip (tiles 309 / 310: weights HBM -> registers, D k tiles ahead) -- operator tests, then the HBM-cold microbench of the M = 128 shapes
set -x
OUT=gpurun_out/${OUTDIR:-r05x}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_planes_gpu.py -m gpu -x -q -k "register_streamed" > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt | cut -c1-400
timeout 600 python - > $OUT/cold_m128.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
sd.set_option("bench_cold", 1)
CASES = [(2, 1280, 8, 8, 1280, 3), (2, 2560, 8, 8, 1280, 3), (2, 1280, 8, 8, 1280, 1), (2, 2560, 8, 8, 1280, 1), (1, 5120, 1, 128, 1280, 1), (1, 1280, 1, 128, 3840, 1), (2, 1280, 16, 16, 1280, 3), (2, 640, 16, 16, 1280, 3)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    wbytes = cout * cin * k * k * 6
    row = f"cold n={n} cin={cin} {h}x{w} cout={cout} k={k} ({wbytes / 1e6:.0f} MB of planes):"
    for fam, tiles in (("LDS tiles", (300, 301, 303, 304, 305, 308)), ("309", (309,)), ("310", (310,))):
        res = []
        for t in tiles:
            for sp in (1, 2, 4, 6, 8, 12, 16, 24, 25, 32, 45, 48, 51, 64):
                try:
                    ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=t, splitk=sp, iters=4)
                except Exception:
                    continue
                res.append((ms, t, sp))
        res.sort()
        ms, t, sp = res[0]
        row += f"  {fam}: {ms * 1e3:6.1f} us (tile {t}, splitk {sp}; weights at {wbytes / ms / 1e9:5.2f} TB/s)"
        if fam != "LDS tiles":
            row += " [" + " ".join(f"{s}:{m * 1e3:.0f}" for m, _, s in sorted(res, key=lambda r: r[2])) + "]"
    print(row, flush=True)
PY
grep "^cold" $OUT/cold_m128.txt | cut -c1-600
