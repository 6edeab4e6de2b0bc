#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
timeout 300 python tools/probes/cold_vs_hot.py stable_diffusion_burn_amd/tuning/gfx950_fp32_planes.txt > $out/cold_vs_hot.txt 2>&1
echo "rc=$?"; cat $out/cold_vs_hot.txt
