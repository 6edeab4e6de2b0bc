#!/bin/bash
# round 3, GPU call h: the tile tables re-measured with the weights evicted between launches (bench_cold = 1: what a layer sees inside the model), then the headline A/B
out=gpurun_out/r03h; mkdir -p $out
T=stable_diffusion_burn_amd/tuning
timeout 420 python tools/autotune.py --shapes-file tools/probes/shapes_b1.txt --families p --iters 4 --budget-s 360 --opt bench_cold=1 --out $out/tune_planes_cold.json --emit $out/gfx950_fp32_planes_cold.txt > $out/tune_planes_cold.log 2>&1
echo "autotune p rc=$?"; tail -3 $out/tune_planes_cold.log
timeout 560 python tools/autotune.py --shapes-file tools/probes/shapes_b1.txt --families old,x,s --iters 4 --budget-s 500 --opt bench_cold=1 --merge $T/gfx950_fp32.txt --out $out/tune_fp32_cold.json --emit $out/gfx950_fp32_cold.txt > $out/tune_fp32_cold.log 2>&1
echo "autotune s rc=$?"; tail -3 $out/tune_fp32_cold.log
timeout 400 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_cold_tables.jsonl \
   --arms gemm_planes=0,tunefile=$T/gfx950_fp32.txt gemm_planes=0,tunefile=$out/gfx950_fp32_cold.txt gemm_planes=1,tunefile=$T/gfx950_fp32_planes.txt gemm_planes=1,tunefile=$out/gfx950_fp32_planes_cold.txt > $out/ab.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_cold_tables.jsonl 2>/dev/null | cut -c1-640; tail -3 $out/ab.log
