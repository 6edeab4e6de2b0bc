#!/bin/bash
# round 3, first GPU call (prepared at the end of round 2): does k_gemm3y.hip (the split GEMM on v_mfma_f32_32x32x16_bf16, 32 x 160 wave tiles) compute
# the right thing, and is it faster?  1. its gated parity tests; 2. the headline with the built-in tile table vs the same table on the 32x32 tiles.
R=$PWD; out=gpurun_out/r03a; mkdir -p $out
export SDMI_UNVERIFIED=1
timeout 240 python -m pytest tests -m "gpu and unverified" -q -p no:cacheprovider -n 6 > $out/pytest_unverified.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest_unverified.log
tail -6 $out/pytest_unverified.log
T=stable_diffusion_burn_amd/tuning
timeout 150 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_gemm3y.jsonl \
    --arms tunefile=$T/gfx950_fp32.txt tunefile=$T/gfx950_fp32_y.txt > $out/ab_fp32_b1_gemm3y.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_gemm3y.jsonl 2>/dev/null | cut -c1-420
