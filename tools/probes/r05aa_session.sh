#!/bin/bash
# round 5, probe aa: fp32 GroupNorm as one launch (gn32_fused): operator tests, golden fp32 fixtures, per-image A/B interleaved in one process
set -x
OUT=gpurun_out/${OUTDIR:-r05aa}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "group_norm" > $OUT/tests_ops.txt 2>&1; tail -3 $OUT/tests_ops.txt | cut -c1-300
timeout 300 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "unet_forward_full or config1_one_step or config2_20_steps_cfg or unpadded_contexts_full_size_fp32" > $OUT/tests_golden.txt 2>&1; tail -2 $OUT/tests_golden.txt | cut -c1-300
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "gn32_fused=0" "gn32_fused=1" > $OUT/ab_fp32_b1.txt 2>&1; grep '^{' $OUT/ab_fp32_b1.txt | cut -c1-520
