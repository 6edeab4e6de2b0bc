#!/bin/bash
# round 5, probe w: weight planes in 16-row fragment groups (b3_grouped, 1 KiB DMA pieces) against row-major planes (64-byte pieces 6 K bytes apart):
# HBM-cold microbench of the weight-stream-bound shapes, the fp32 GPU tests, and the batch-1 image in two processes on this box
set -x
OUT=gpurun_out/${OUTDIR:-r05w}; mkdir -p $OUT
timeout 900 python - > $OUT/cold_weight_bound.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sds = {}
for g in (1, 0):
    sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
    sd.set_option("b3_grouped", g)
    sd.set_option("bench_cold", 1)
    sds[g] = sd
CASES = [(2, 1280, 8, 8, 1280, 3), (2, 2560, 8, 8, 1280, 3), (2, 1280, 16, 16, 1280, 3), (2, 2560, 16, 16, 1280, 3), (2, 1920, 16, 16, 1280, 3), (2, 1280, 16, 16, 1280, 1), (2, 1280, 8, 8, 1280, 1),
         (2, 640, 32, 32, 640, 3), (2, 1280, 32, 32, 640, 3), (2, 320, 64, 64, 320, 3), (1, 1280, 1, 128, 10240, 1), (1, 5120, 1, 128, 1280, 1)]
for (n, cin, h, w, cout, k) in CASES:
    fl = 2.0 * n * h * w * cout * cin * k * k
    wbytes = cout * cin * k * k * 6
    row = f"cold n={n} cin={cin} {h}x{w} cout={cout} k={k} ({wbytes / 1e6:.0f} MB of planes):"
    for g in (0, 1, 0, 1):
        best = None
        for t in (300, 301, 303, 304, 305, 308):
            for sp in (1, 2, 4, 8, 16, 32):
                try:
                    ms = sds[g].bench_conv(n, cin, h, w, cout, k=k, tile_cfg=t, splitk=sp, iters=4)
                except Exception:
                    continue
                if best is None or ms < best[0]:
                    best = (ms, t, sp)
        ms, t, sp = best
        row += f"  {'grouped' if g else 'row-major'}: {ms * 1e3:7.1f} us (tile {t}, splitk {sp}; {fl / ms / 1e9:5.0f} TF, weights at {wbytes / ms / 1e9:5.2f} TB/s)"
    print(row, flush=True)
PY
echo "rc=$?"; grep "^cold" $OUT/cold_weight_bound.txt | cut -c1-520
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py tests/test_model_gpu.py -m gpu -x -q > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "unet_forward_full or config1_one_step or config2_20_steps_cfg or unpadded_contexts_full_size_fp32" > $OUT/tests_golden.txt 2>&1; tail -3 $OUT/tests_golden.txt
for rep in 1 2; do
SDMI_OPTS="b3_grouped=0" timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "attn_pack_tail=3" > $OUT/ab_rowmajor_$rep.txt 2>&1; grep '^{' $OUT/ab_rowmajor_$rep.txt | cut -c1-420
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "attn_pack_tail=3" > $OUT/ab_grouped_$rep.txt 2>&1; grep '^{' $OUT/ab_grouped_$rep.txt | cut -c1-420
done
