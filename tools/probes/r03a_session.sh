#!/bin/bash
# round 3, first GPU call (prepared at the end of round 2): do k_gemm3y.hip / k_gemm_bf16y.hip (the split and the bf16 GEMM on v_mfma_f32_32x32x16_bf16) compute
# the right thing, and is it faster?  1. its gated parity tests; 2. the headline (and the bf16 batch-8 shard) with option gemm_y = 0 / 1: every large-tile launch on the 32x32x16 tile of the same shape.
R=$PWD; out=gpurun_out/r03a; mkdir -p $out
export SDMI_UNVERIFIED=1
timeout 240 python -m pytest tests -m "gpu and unverified" -q -p no:cacheprovider -n 6 > $out/pytest_unverified.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest_unverified.log
tail -6 $out/pytest_unverified.log
timeout 150 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_gemm_y.jsonl \
    --arms gemm_y=0 gemm_y=1 > $out/ab_fp32_b1_gemm_y.log 2>&1
echo "ab fp32 rc=$?"; cat $out/ab_fp32_b1_gemm_y.jsonl 2>/dev/null | cut -c1-420
timeout 150 python tools/ab_variants.py --precision bf16 --batch 8 --rounds 2 --out $out/ab_bf16_b8_gemm_y.jsonl \
    --arms gemm_y=0 gemm_y=1 > $out/ab_bf16_b8_gemm_y.log 2>&1
echo "ab bf16 rc=$?"; cat $out/ab_bf16_b8_gemm_y.jsonl 2>/dev/null | cut -c1-420
