#!/bin/bash
# round 3, GPU call r (the last minutes of the budget): the model-level tests whose bars / helper code changed after call q, serially
out=gpurun_out/r03r; mkdir -p $out
timeout 70 python -m pytest tests/test_golden_gpu.py -q -p no:cacheprovider -x -k "config5 or config3_bf16" > $out/pytest_golden.log 2>&1; echo "golden rc=$?"; tail -3 $out/pytest_golden.log | cut -c1-200
timeout 200 python -m pytest tests/test_fp8_gpu.py -q -p no:cacheprovider -x --durations=8 > $out/pytest_fp8.log 2>&1; echo "fp8 rc=$?"; tail -14 $out/pytest_fp8.log | cut -c1-200
timeout 130 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x --durations=5 -k "unet_forward or sample_image or batch_and" > $out/pytest_bf16.log 2>&1; echo "bf16 rc=$?"; tail -10 $out/pytest_bf16.log | cut -c1-200
