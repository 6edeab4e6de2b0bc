#!/bin/bash
# round 2, last GPU minutes: parity of the pipelined k loops (options, off by default) and their A/B against the default loops.
# Everything lands in gpurun_out/r02y/ as it is produced (the call may be cut at its limit).
R=$PWD; out=gpurun_out/r02y; mkdir -p $out
export SDMI_UNVERIFIED=1
date +%s > $out/t0
# 1. parity / bit-identity of the new variants, six workers (each with its own engines)
timeout 200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_bf16_gpu.py -m "gpu and unverified" -q -p no:cacheprovider -n 6 > $out/pytest_unverified.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest_unverified.log
tail -5 $out/pytest_unverified.log
date +%s > $out/t1
# 2. fp32 batch 1 (the headline): default loop vs the hoisted ones
timeout 150 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1.jsonl \
    --arms gemm3x_variant=2 gemm3x_variant=10 gemm3x_variant=42 gemm3x_variant=58 gemm3x_variant=46 > $out/ab_fp32_b1.log 2>&1
echo "ab fp32 rc=$?"; cat $out/ab_fp32_b1.jsonl 2>/dev/null | cut -c1-400
date +%s > $out/t2
# 3. bf16 batch 8 (configs[3] shard): plain vs pipelined loop
timeout 150 python tools/ab_variants.py --precision bf16 --batch 8 --rounds 2 --out $out/ab_bf16_b8.jsonl \
    --arms gemm_bf16x_variant=0 gemm_bf16x_variant=1 > $out/ab_bf16_b8.log 2>&1
echo "ab bf16 rc=$?"; cat $out/ab_bf16_b8.jsonl 2>/dev/null | cut -c1-400
date +%s > $out/t3
