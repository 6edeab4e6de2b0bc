#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -n 6 -x > $out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 $out/pytest_gpu.log
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_fuse.jsonl \
   --arms gemm_planes=0,fuse_reduce=0 gemm_planes=0,fuse_reduce=1 gemm_planes=1,fuse_reduce=1 > $out/ab1.log 2>&1
echo "ab1 rc=$?"; cat $out/ab_fp32_b1_fuse.jsonl 2>/dev/null | cut -c1-640; tail -2 $out/ab1.log
timeout 300 python tools/ab_variants.py --precision fp32 --batch 4 --rounds 2 --out $out/ab_fp32_b4_planes.jsonl \
   --arms gemm_planes=0 gemm_planes=1 > $out/ab4.log 2>&1
echo "ab4 rc=$?"; cat $out/ab_fp32_b4_planes.jsonl 2>/dev/null | cut -c1-640; tail -2 $out/ab4.log
