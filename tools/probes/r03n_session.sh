#!/bin/bash
# round 3, GPU call n: per-workgroup phase stamps of the plane GEMM (where a batch-1 launch spends its time)
out=gpurun_out/r03n; mkdir -p $out
timeout 300 python tools/probes/gemm_phase_probe.py > $out/probe.out 2> $out/gemm_phase_probe.txt; echo "probe rc=$?"
grep -v amdgpu.ids $out/gemm_phase_probe.txt | cut -c1-420 | head -80
