#!/bin/bash
# round 4, call k: bf16 attention in two wave groups one phase apart (softmax of one wave beside the matrix phase of its SIMD partner)
out=gpurun_out/r04k; mkdir -p $out
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -k "attention" > $out/pytest_attn.log 2>&1; echo "attn tests rc=$?"; tail -4 $out/pytest_attn.log | cut -c1-300
timeout 300 python tools/bench_attn.py --bf16 --b16 > $out/bench_attn_b16.txt 2>&1; grep -v amdgpu.ids $out/bench_attn_b16.txt | cut -c1-200
timeout 300 python tools/bench_attn.py --bf16 > $out/bench_attn_b2.txt 2>&1; grep -v amdgpu.ids $out/bench_attn_b2.txt | cut -c1-200
