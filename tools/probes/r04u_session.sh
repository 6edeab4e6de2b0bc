#!/bin/bash
# round 4, call u: parity of the kernel-row tiles (104 / 105), their A/B against 100 / 101 per shape, and the model with conv3_reuse = 0 / 1
out=gpurun_out/r04u; mkdir -p $out
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q -k "kernel_row or large_tiles" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout 900 python tools/probes/r04u_ab_conv3.py > $out/ab.log 2>&1; echo "ab rc=$?"; cat $out/ab.log
for o in 0 1 0 1; do
  timeout 300 python bench.py --precision bf16 --batch-per-gpu 16 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --opt conv3_reuse=$o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv3_reuse=$o', d['value'], d['unit'], d['ms_per_step'])"
done
