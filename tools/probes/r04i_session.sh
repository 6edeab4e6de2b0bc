#!/bin/bash
# round 4, call i: bf16 attention with one exponential per score (q in log2 units, -m as the accumulator input, deferred rescale, row sum by a ones column)
out=gpurun_out/r04i; mkdir -p $out
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x -k "attention" > $out/pytest_attn.log 2>&1; echo "attn tests rc=$?"; tail -4 $out/pytest_attn.log | cut -c1-300
timeout 300 python tools/bench_attn.py --bf16 --b16 > $out/bench_attn_b16.txt 2>&1; grep -v amdgpu.ids $out/bench_attn_b16.txt | cut -c1-200
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -q -p no:cacheprovider -x > $out/pytest_bf16_fp8.log 2>&1; echo "bf16+fp8 tests rc=$?"; tail -4 $out/pytest_bf16_fp8.log | cut -c1-300
