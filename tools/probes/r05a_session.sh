#!/bin/bash
# round 5, first GPU contact: parity of the persistent tile loop / the direct epilogue (bit-identical to variant 0), the short-K sweep per variant, A/B per image
out=gpurun_out/r05a; mkdir -p $out
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q -k "persistent or large_tiles or kernel_row" 2>&1 | tail -15 > $out/pytest_variants.txt; echo "pytest rc=$?"; tail -5 $out/pytest_variants.txt
timeout 600 python tools/probes/r05a_shortk.py > $out/shortk_variants.txt 2> $out/shortk_variants.err; echo "shortk rc=$?"; cat $out/shortk_variants.txt
timeout 600 python tools/ab_variants.py --precision bf16 --batch 8 --arms "gemm_bf16x_variant=0" "gemm_bf16x_variant=1" "gemm_bf16x_variant=2" "gemm_bf16x_variant=3" --rounds 3 --out $out/ab_bf16_b8.jsonl > $out/ab_bf16_b8.log 2>&1; echo "ab bf16 b8 rc=$?"; cut -c1-600 $out/ab_bf16_b8.jsonl
timeout 600 python tools/ab_variants.py --precision bf16 --batch 16 --arms "gemm_bf16x_variant=0" "gemm_bf16x_variant=3" --rounds 2 --out $out/ab_bf16_b16.jsonl > $out/ab_bf16_b16.log 2>&1; echo "ab bf16 b16 rc=$?"; cut -c1-600 $out/ab_bf16_b16.jsonl
timeout 600 python tools/ab_variants.py --precision fp8 --batch 16 --arms "gemm_bf16x_variant=0" "gemm_bf16x_variant=2" "gemm_bf16x_variant=3" --rounds 2 --out $out/ab_fp8_b16.jsonl > $out/ab_fp8_b16.log 2>&1; echo "ab fp8 b16 rc=$?"; cut -c1-600 $out/ab_fp8_b16.jsonl
