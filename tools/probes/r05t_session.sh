#!/bin/bash
# round 5, probe t: the non-temporal store threshold (option nt_store_mb) per image, interleaved in one process, plus bit-identity of the bf16 tests
set -x
OUT=gpurun_out/r05t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -m gpu -x -q > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
timeout 600 python tools/ab_variants.py --precision bf16 --batch 16 --rounds 3 --arms "nt_store_mb=0" "nt_store_mb=64" "nt_store_mb=128" "nt_store_mb=256" > $OUT/ab_bf16_b16.txt 2>&1; grep '^{' $OUT/ab_bf16_b16.txt | cut -c1-400
timeout 600 python tools/ab_variants.py --precision fp8 --batch 16 --rounds 3 --arms "nt_store_mb=0" "nt_store_mb=64" "nt_store_mb=128" "nt_store_mb=256" > $OUT/ab_fp8_b16.txt 2>&1; grep '^{' $OUT/ab_fp8_b16.txt | cut -c1-400
timeout 600 python tools/ab_variants.py --precision bf16 --batch 8 --rounds 3 --arms "nt_store_mb=0" "nt_store_mb=64" "nt_store_mb=128" > $OUT/ab_bf16_b8.txt 2>&1; grep '^{' $OUT/ab_bf16_b8.txt | cut -c1-400
