#!/bin/bash
# round 3, GPU call c: producers writing planes (option gemm_planes = 1) -- parity, the plane-tile table, and the headline A/B
out=gpurun_out/r03c; mkdir -p $out
timeout 500 python -m pytest tests/test_planes_gpu.py -q -p no:cacheprovider -n 4 > $out/pytest_planes.log 2>&1
echo "pytest rc=$?"; tail -12 $out/pytest_planes.log
timeout 120 python tools/record_shapes.py $out/shapes_b1.txt > $out/record_shapes.log 2>&1
echo "record rc=$?"; wc -l $out/shapes_b1.txt
timeout 420 python tools/autotune.py --shapes-file $out/shapes_b1.txt --families p --iters 5 --budget-s 360 --out $out/tune_planes.json --emit $out/gfx950_fp32_planes.txt > $out/tune_planes.log 2>&1
echo "autotune rc=$?"; tail -4 $out/tune_planes.log; wc -l $out/gfx950_fp32_planes.txt
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1_planes.jsonl \
   --arms gemm_planes=0 gemm_planes=1 gemm_planes=1,tunefile=$out/gfx950_fp32_planes.txt > $out/ab_fp32_b1_planes.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp32_b1_planes.jsonl 2>/dev/null | cut -c1-700; tail -5 $out/ab_fp32_b1_planes.log
