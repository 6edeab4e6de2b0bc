#!/bin/bash
# round-2 session h: validate k_gemm3x.hip (fp32 on the bf16 matrix pipe), pick the DMA variant, tune, bench
out=gpurun_out/r02h; mkdir -p $out
timeout 420 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "split_bf16_is or small_integers" 2>&1 | tail -8 > $out/tests.txt; cat $out/tests.txt
for v in 0 1; do
  timeout 120 python tools/autotune.py --quick --families s --opt gemm3x_variant=$v --out $out/quick_v$v.json --budget-s 60 > $out/quick_v$v.txt 2>&1; tail -12 $out/quick_v$v.txt
done
timeout 420 python tools/autotune.py --shapes-file profiles/r01_gemm_shapes_b1.txt --families s --merge stable_diffusion_burn_amd/tuning/gfx950_fp32.txt \
   --emit $out/gfx950_fp32.txt --out $out/tune_fp32_s.json --budget-s 330 > $out/tune.txt 2>&1; tail -5 $out/tune.txt
timeout 200 python bench.py --no-secondary --no-cpu-baseline --tune-file $out/gfx950_fp32.txt > $out/bench_s.json 2> $out/bench_s.err; tail -c 1500 $out/bench_s.json
