#!/bin/bash
# round 6, call y: staggered DMA issue (gemm_bf16x_variant bit 2: waves 4-7 issue between a tile's two k steps) in the kernel-row convolution (k_gemm_bf16t.hip) and the one-tile
# 256 x 256 / 256 x 128 forms: bit-identity tests, per shape (engine's choice, hot / cold), per image
out=gpurun_out
SDMI_OPTS="gemm_bf16x_variant=5" python -m pytest tests/test_bf16_gpu.py -x -q -k "conv or linear or kernel_row or persistent" > $out/r06y_pytest_variant5.txt 2>&1; tail -n 2 $out/r06y_pytest_variant5.txt
python - > $out/r06y_stagger_shapes.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, '.')
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
SH = [((32, 320, 64, 64, 320), 3), ((32, 640, 64, 64, 320), 3), ((32, 960, 64, 64, 320), 3), ((32, 640, 32, 32, 640), 3), ((32, 1280, 32, 32, 640), 3), ((32, 1920, 32, 32, 640), 3),
      ((32, 1280, 16, 16, 1280), 3), ((32, 2560, 16, 16, 1280), 3), ((1, 512, 128, 128, 512), 3), ((1, 256, 256, 256, 256), 3), ((1, 128, 512, 512, 128), 3)]
for cold in (0, 1):
    sd.set_option("bench_cold", cold)
    for shape, k in SH:
        r = []
        for v in (1, 5, 1, 5):
            sd.set_option("gemm_bf16x_variant", v)
            r.append(sd.bench_conv(*shape, k=k, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4) * 1e3)
        print(("cold " if cold else "hot  ") + f"{str(shape):30s} k{k}  engine's tile: {r[0]:7.1f} {r[2]:7.1f} -> stagger {r[1]:7.1f} {r[3]:7.1f}", flush=True)
PY
cat $out/r06y_stagger_shapes.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 --rounds 3 --out $out/r06y_ab_stagger_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 --rounds 3 --out $out/r06y_ab_stagger_fp8_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision bf16 --batch 8 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 --rounds 3 --out $out/r06y_ab_stagger_b8.jsonl > /dev/null 2>&1
cat $out/r06y_ab_stagger_*.jsonl | cut -c1-330
