#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
T=stable_diffusion_burn_amd/tuning
timeout 400 python -m pytest tests/test_planes_gpu.py -q -p no:cacheprovider -n 6 -x > $out/pytest_planes.log 2>&1
echo "pytest rc=$?"; tail -6 $out/pytest_planes.log | cut -c1-300
timeout 460 python tools/autotune.py --shapes-file tools/probes/shapes_b1.txt --families p --iters 4 --budget-s 400 --opt bench_cold=1 --out $out/tune_planes_cold2.json --emit $out/gfx950_fp32_planes_cold2.txt > $out/tune_planes_cold2.log 2>&1
echo "autotune p rc=$?"; tail -3 $out/tune_planes_cold2.log; grep -c "=30[5-8]," $out/gfx950_fp32_planes_cold2.txt
timeout 400 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1.jsonl \
   --arms gemm_planes=0,fuse_reduce=0 gemm_planes=1,fuse_reduce=0,tunefile=$out/gfx950_fp32_planes_cold2.txt gemm_planes=1,fuse_reduce=1,tunefile=$out/gfx950_fp32_planes_cold2.txt gemm_planes=0,fuse_reduce=1 > $out/ab.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1.jsonl 2>/dev/null | cut -c1-640; tail -3 $out/ab.log
