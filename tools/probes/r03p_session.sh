#!/bin/bash
# round 3, GPU calls o / p (history: commit "Split-K combined inside the launch ..."): the in-launch split-K combine -- agreement with the separate
# reduce launch (tools/probes/csk_check.py of that commit), per-phase stamps (gemm_phase_probe.py 1) and the end-to-end A/B.  Results:
# profiles/r03o_*, profiles/r03p_*.  The code was removed again (3.5 % slower); this script is kept as the record of what was run.
out=gpurun_out/r03p; mkdir -p $out
timeout 300 python tools/probes/csk_check.py > $out/csk_check.txt 2>&1; echo "csk_check rc=$?"
timeout 300 python tools/probes/gemm_phase_probe.py 1 > $out/probe.out 2> $out/gemm_phase_probe_coop.txt; echo "probe rc=$?"
timeout 300 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 2 --out $out/ab_fp32_b1_splitk_coop.jsonl --arms splitk_coop=0 splitk_coop=1 > $out/ab.log 2>&1
