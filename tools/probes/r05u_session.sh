#!/bin/bash
# round 5, probe u: fp32 attention at d = 40 with the packed tail (attn_pack_tail): operator tests, the fp32 golden fixtures, per-image A/B interleaved in one process
set -x
OUT=gpurun_out/${OUTDIR:-r05u}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" -s > $OUT/tests_ops.txt 2>&1; tail -3 $OUT/tests_ops.txt; grep "^attention" $OUT/tests_ops.txt | head -30
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "unet_forward_full or config1_one_step or config2_20_steps_cfg or unpadded_contexts_full_size_fp32" > $OUT/tests_golden.txt 2>&1; tail -3 $OUT/tests_golden.txt
timeout 600 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "attn_pack_tail=1" "attn_pack_tail=3" "attn_pack_tail=0" > $OUT/ab_fp32_b1.txt 2>&1; grep '^{' $OUT/ab_fp32_b1.txt | cut -c1-600
