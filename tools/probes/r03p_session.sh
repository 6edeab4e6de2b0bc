#!/bin/bash
out=gpurun_out/r03p; mkdir -p $out
timeout 300 python tools/probes/csk_check.py > $out/csk_check.txt 2>&1; echo "csk_check rc=$?"; grep -v amdgpu.ids $out/csk_check.txt | tail -18 | cut -c1-200
timeout 300 python tools/probes/gemm_phase_probe.py 1 > $out/probe.out 2> $out/gemm_phase_probe_coop.txt; echo "probe rc=$?"
grep -v amdgpu.ids $out/gemm_phase_probe_coop.txt | grep "in-launch\|timed" | cut -c1-330 | head -60
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 2 --out $out/ab_fp32_b1_splitk_coop.jsonl --arms splitk_coop=0 splitk_coop=1 > $out/ab.log 2>&1
echo "ab rc=$?"; cut -c1-700 $out/ab_fp32_b1_splitk_coop.jsonl
