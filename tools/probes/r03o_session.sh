#!/bin/bash
# round 3, GPU call o: the in-launch split-K combine -- agreement with the separate reduce launch, then the end-to-end A/B
out=gpurun_out/r03o; mkdir -p $out
timeout 300 python tools/probes/csk_check.py > $out/csk_check.txt 2>&1; echo "csk_check rc=$?"; grep -v amdgpu.ids $out/csk_check.txt | tail -25
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 2 --out $out/ab_fp32_b1_splitk_coop.jsonl --arms splitk_coop=0 splitk_coop=1 > $out/ab.log 2>&1
echo "ab rc=$?"; cut -c1-700 $out/ab_fp32_b1_splitk_coop.jsonl; tail -3 $out/ab.log | cut -c1-300
