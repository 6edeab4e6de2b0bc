#!/bin/bash
# (record only: the temporary switches / variant this call measured are not in the tree; see profiles/README.md, "Round 4")
# round 4, call aa: the erf-GELU of the reduced-precision paths by A&S 7.1.26 (13 instructions) instead of erff (~30): parity of the bf16 / precision-2 suites, then the models
out=gpurun_out/r04aa; mkdir -p $out
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -q -p no:cacheprovider -x > $out/pytest_bf16_fp8.log 2>&1; echo "bf16+fp8 tests rc=$?"; tail -3 $out/pytest_bf16_fp8.log | cut -c1-200
timeout 1500 python -m pytest tests/test_golden_gpu.py -q -p no:cacheprovider -x -s -k "bf16 or mxfp8 or reduced" > $out/pytest_golden.log 2>&1; echo "golden rc=$?"; grep -i "rel-RMS\|passed\|failed" $out/pytest_golden.log | cut -c1-220
for i in 1 2; do
  timeout 300 python bench.py --precision bf16 --batch-per-gpu 16 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 b16 s20', d['value'], d['unit'], d['ms_per_step'])"
done
timeout 300 python bench.py --precision fp8 --batch-per-gpu 16 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp8 b16 s20', d['value'], d['unit'], d['ms_per_step'])"
