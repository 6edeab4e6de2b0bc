#!/bin/bash
# round 6, call zw: one-launch GroupNorm for small fp32 tensors (k_norm.hip gn_fused_kernel; engine option gn32_fused) -- GPU tests that touch it, interleaved A/B at fp32 B = 1
out=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py tests/test_golden_gpu.py tests/test_model_gpu.py -x -q -m gpu > $out/r06zw_pytest.txt 2>&1; grep -n "passed\|failed" $out/r06zw_pytest.txt | tail -n 2
python tools/ab_variants.py --precision fp32 --batch 1 --arms gn32_fused=0 gn32_fused=1 --rounds 4 --out $out/r06zw_ab_fp32_b1.jsonl > $out/r06zw_ab.log 2>&1
cut -c1-450 $out/r06zw_ab_fp32_b1.jsonl
