#!/bin/bash
# round 5, probe ah: the fp32 plane GEMM's fused GEGLU gate with gelu_gate_fast (gemm3x_variant bit 5) against erff(), per image, interleaved
OUT=gpurun_out/r05ah; mkdir -p $OUT
timeout 120 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "gemm3x_variant=2" "gemm3x_variant=34" > $OUT/ab_fp32_b1.txt 2>&1; grep '^{' $OUT/ab_fp32_b1.txt | cut -c1-420
