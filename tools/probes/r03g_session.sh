#!/bin/bash
out=gpurun_out/r03g; mkdir -p $out
timeout 400 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_prefetch_plain.jsonl \
   --arms gemm_planes=0,weight_prefetch=0 gemm_planes=0,weight_prefetch=32 gemm_planes=0,weight_prefetch=128 gemm_planes=1,weight_prefetch=32 gemm_planes=1,weight_prefetch=256 > $out/ab.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_prefetch_plain.jsonl 2>/dev/null | cut -c1-560; tail -3 $out/ab.log
