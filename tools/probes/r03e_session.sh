#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
timeout 300 python tools/probes/cold_vs_hot.py stable_diffusion_burn_amd/tuning/gfx950_fp32_planes.txt touch > $out/cold_vs_hot_touch.txt 2>&1
echo "rc=$?"; cat $out/cold_vs_hot_touch.txt
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_prefetch.jsonl \
   --arms gemm_planes=0,weight_prefetch=0 gemm_planes=0,weight_prefetch=1 gemm_planes=1,weight_prefetch=0 gemm_planes=1,weight_prefetch=1 > $out/ab_fp32_b1_prefetch.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_prefetch.jsonl 2>/dev/null | cut -c1-700; tail -3 $out/ab_fp32_b1_prefetch.log
