#!/bin/bash
# round 3, GPU call l: precision = 2 widened to the Linear layers and the 1x1 / up / down convolutions (option fp8_linear) -- parity at operator,
# model and full size (configs[4] batch 16), then the A/B
out=gpurun_out/r03l; mkdir -p $out
timeout 420 python -m pytest tests/test_fp8_gpu.py -q -p no:cacheprovider -n 8 -s > $out/pytest_fp8.log 2>&1
echo "pytest fp8 rc=$?"; grep -E "passed|failed|rel-RMS|identical|kernels per|Error|error" $out/pytest_fp8.log | cut -c1-400 | tail -30
timeout 420 python -m pytest tests/test_golden_gpu.py -k config5 -q -p no:cacheprovider -n 2 -s > $out/pytest_cfg5.log 2>&1
echo "pytest cfg5 rc=$?"; grep -E "passed|failed|rel-RMS|Error|error" $out/pytest_cfg5.log | cut -c1-400 | tail -20
timeout 300 python tools/ab_variants.py --precision fp8 --batch 16 --rounds 2 --out $out/ab_fp8_b16_linear.jsonl --arms fp8_linear=0 fp8_linear=1 > $out/ab.log 2>&1
echo "ab rc=$?"; cat $out/ab_fp8_b16_linear.jsonl 2>/dev/null | cut -c1-600; tail -3 $out/ab.log
