#!/bin/bash
# round 2: the XCD-aware work map of the GEMM launches (option xcd_map): bit-identity tests, then A/B against the legacy map.
R=$PWD; out=gpurun_out/r02z; mkdir -p $out
export SDMI_UNVERIFIED=1
timeout 150 python -m pytest tests -m "gpu and unverified" -q -p no:cacheprovider -n 4 > $out/pytest_unverified.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest_unverified.log
tail -4 $out/pytest_unverified.log
timeout 120 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1.jsonl \
    --arms xcd_map=0,gemm3x_variant=2 xcd_map=1,gemm3x_variant=2 xcd_map=1,gemm3x_variant=42 > $out/ab_fp32_b1.log 2>&1
echo "ab fp32 rc=$?"; cat $out/ab_fp32_b1.jsonl 2>/dev/null | cut -c1-420
timeout 120 python tools/ab_variants.py --precision bf16 --batch 8 --rounds 2 --out $out/ab_bf16_b8.jsonl \
    --arms xcd_map=0 xcd_map=1 > $out/ab_bf16_b8.log 2>&1
echo "ab bf16 rc=$?"; cat $out/ab_bf16_b8.jsonl 2>/dev/null | cut -c1-420
