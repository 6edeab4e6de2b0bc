#!/bin/bash
# round 6, call h: (i) bias + time-embedding row as accumulator input too (resid_acc=3 against 1) in the reduced precisions; (ii) fp32: the epilogue's loads issued early
# (gemm3x_variant bit 5 = off) at batch 1; operator tests of both
out=gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06h_pytest_bf16_fp8.txt 2>&1
tail -n 3 $out/r06h_pytest_bf16_fp8.txt
python -m pytest tests/test_ops_gpu.py tests/test_planes_gpu.py -x -q > $out/r06h_pytest_ops_planes.txt 2>&1
tail -n 3 $out/r06h_pytest_ops_planes.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms resid_acc=1 resid_acc=3 --rounds 3 --out $out/r06h_ab_bias_acc_b16.jsonl > $out/r06h_ab1.log 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms resid_acc=1 resid_acc=3 --rounds 3 --out $out/r06h_ab_bias_acc_fp8_b16.jsonl > $out/r06h_ab2.log 2>&1
python tools/ab_variants.py --precision fp32 --batch 1 --arms gemm3x_variant=34 gemm3x_variant=2 --rounds 4 --out $out/r06h_ab_fp32_epilogue_early_loads.jsonl > $out/r06h_ab3.log 2>&1
python -m pytest tests/test_golden_gpu.py -x -q > $out/r06h_pytest_golden.txt 2>&1
tail -n 3 $out/r06h_pytest_golden.txt
cat $out/r06h_ab_*.jsonl | cut -c1-420
