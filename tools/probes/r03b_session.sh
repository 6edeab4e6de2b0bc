#!/bin/bash
# round 3, GPU call b: k_gemm3p.hip (activations as bf16 planes too) -- parity, per-shape time against k_gemm3x.hip, and the headline with every split-GEMM launch
# moved to the plane kernel (the activations still converted by split3_rows_kernel in front of every launch: its cost is the class "split_rows")
out=gpurun_out/r03b; mkdir -p $out
timeout 400 python -m pytest tests/test_planes_gpu.py -q -p no:cacheprovider -n 4 > $out/pytest_planes.log 2>&1
echo "pytest rc=$?"; tail -15 $out/pytest_planes.log
timeout 260 python tools/autotune.py --quick --families s,p --iters 5 --budget-s 200 --out $out/tune_quick_sp.json > $out/tune_quick_sp.log 2>&1
echo "autotune rc=$?"; tail -12 $out/tune_quick_sp.log
timeout 200 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_planes.jsonl --arms gemm_planes=0 gemm_planes=1 > $out/ab_fp32_b1_planes.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_planes.jsonl 2>/dev/null | cut -c1-600; tail -5 $out/ab_fp32_b1_planes.log
