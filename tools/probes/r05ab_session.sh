#!/bin/bash
# round 5, probe ab: non-temporal LDS-DMA for the weight pieces of single-M-tile layers (gemm3x_variant bit 6), per image, interleaved
set -x
OUT=gpurun_out/r05ab; mkdir -p $OUT
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "gemm3x_variant=2" "gemm3x_variant=66" > $OUT/ab_fp32_b1.txt 2>&1; grep '^{' $OUT/ab_fp32_b1.txt | cut -c1-520
timeout 200 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "config1_one_step" > $OUT/t.txt 2>&1; tail -1 $OUT/t.txt
