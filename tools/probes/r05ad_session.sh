#!/bin/bash
# round 5, probe ad: the GEGLU gate's GELU without erff() in the reduced-precision epilogue (gemm_bf16x_variant bit 5 = erff()): operator tests, per image interleaved
set -x
OUT=gpurun_out/r05ad; mkdir -p $OUT
timeout 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "geglu" > $OUT/tests.txt 2>&1; tail -2 $OUT/tests.txt | cut -c1-300
timeout 300 python tools/ab_variants.py --precision bf16 --batch 16 --rounds 3 --arms "gemm_bf16x_variant=33" "gemm_bf16x_variant=1" > $OUT/ab_bf16_b16.txt 2>&1; grep '^{' $OUT/ab_bf16_b16.txt | cut -c1-420
timeout 300 python tools/ab_variants.py --precision fp8 --batch 16 --rounds 3 --arms "gemm_bf16x_variant=33" "gemm_bf16x_variant=1" > $OUT/ab_fp8_b16.txt 2>&1; grep '^{' $OUT/ab_fp8_b16.txt | cut -c1-420
