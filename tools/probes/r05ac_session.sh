#!/bin/bash
# round 5, probe ac: static s_setprio 1 for waves NW/2.. of the attention kernels (attn_pack_tail bit 2), per image, interleaved
set -x
OUT=gpurun_out/r05ac; mkdir -p $OUT
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 4 --arms "attn_pack_tail=3" "attn_pack_tail=7" > $OUT/ab_fp32_b1.txt 2>&1; grep '^{' $OUT/ab_fp32_b1.txt | cut -c1-420
timeout 300 python tools/ab_variants.py --precision bf16 --batch 16 --rounds 3 --arms "attn_pack_tail=3" "attn_pack_tail=7" > $OUT/ab_bf16_b16.txt 2>&1; grep '^{' $OUT/ab_bf16_b16.txt | cut -c1-420
