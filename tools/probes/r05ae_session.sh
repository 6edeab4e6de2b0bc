#!/bin/bash
# round 5, session ae: the reduced-precision GEGLU gate without erff() as the only form -- the operator tests that touch it and the full-size reduced-precision golden fixtures
set -x
OUT=gpurun_out/r05ae; mkdir -p $OUT
timeout 300 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -k "geglu" -s > $OUT/tests_ops.txt 2>&1; tail -2 $OUT/tests_ops.txt | cut -c1-300; grep "geglu_fp8" $OUT/tests_ops.txt | head -3
timeout 600 python -m pytest tests/test_golden_gpu.py -m gpu -x -q -k "config4_shard_bf16 or config5_mxfp8 or bf16_full_size_against or unpadded_contexts_full_size_reduced" -s > $OUT/tests_golden.txt 2>&1; tail -2 $OUT/tests_golden.txt | cut -c1-300
