#!/bin/bash
# round 4, call m: fp32 batch 1, the weight-stream-bound levels (8x8, 16x16): plane tiles (6 bytes per weight) against the fp32-weight kernels (4 bytes), operands HBM-cold
out=gpurun_out/r04m; mkdir -p $out
timeout 900 python - > $out/cold_weight_bound.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
sd.set_option("bench_cold", 1)
CASES = [(2, 1280, 8, 8, 1280, 3), (2, 2560, 8, 8, 1280, 3), (2, 1280, 16, 16, 1280, 3), (2, 2560, 16, 16, 1280, 3), (2, 1920, 16, 16, 1280, 3), (2, 1280, 16, 16, 1280, 1), (2, 1280, 8, 8, 1280, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    wbytes = cout * cin * k * k
    best = {}
    for fam, tiles in (("planes 6B", (300, 301, 303, 304, 305, 306, 307, 308)), ("split 6B w / fp32 act", (200, 202, 204, 205)), ("fp32 mfma 4B", (0, 1, 2, 3, 4, 5, 6, 7, 8, 9))):
        for t in tiles:
            for sp in (1, 2, 4, 8, 16, 32):
                try:
                    ms = sd.bench_conv(n, cin, h, w, cout, k=k, tile_cfg=t, splitk=sp, iters=4)
                except Exception:
                    continue
                if fam not in best or ms < best[fam][0]:
                    best[fam] = (ms, t, sp)
    row = f"cold n={n} cin={cin} {h}x{w} cout={cout} k={k} ({wbytes * 4 / 1e6:.0f} MB fp32 weights):"
    for fam, (ms, t, sp) in best.items():
        row += f"  {fam}: {ms * 1e3:7.1f} us (tile {t}, splitk {sp}; {fl / ms / 1e9:5.0f} TF, weights at {wbytes * (6 if '6B' in fam else 4) / ms / 1e9:5.2f} TB/s)"
    print(row, flush=True)
sd.close()
PY
echo "rc=$?"; grep -v amdgpu.ids $out/cold_weight_bound.txt | cut -c1-420
