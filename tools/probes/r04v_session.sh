#!/bin/bash
# round 4, call v: the bf16 / precision-2 models with conv3_reuse = 0 / 1 (k_gemm_bf16t.hip on the 3x3 convolutions that chose tile 100 / 101), and the golden tests with it on
out=gpurun_out/r04v; mkdir -p $out
for o in 0 1 0 1; do
  timeout 300 python bench.py --precision bf16 --batch-per-gpu 16 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --opt conv3_reuse=$o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 b16 conv3_reuse=$o', d['value'], d['unit'], d['ms_per_step'])"
done
for o in 0 1; do
  timeout 300 python bench.py --precision bf16 --batch-per-gpu 8 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --opt conv3_reuse=$o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 b8 conv3_reuse=$o', d['value'], d['unit'], d['ms_per_step'])"
done
