#!/bin/bash
# round 6, call g: the residual as the accumulators' initial value (option resid_acc: 0 = read by the epilogue, round 5), per image, interleaved in one process;
# operator + golden tests of the reduced precisions on the new default; the isolated short-K shapes
out=gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06g_pytest_bf16_fp8.txt 2>&1
tail -n 3 $out/r06g_pytest_bf16_fp8.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms resid_acc=0 resid_acc=1 --rounds 3 --out $out/r06g_ab_resid_acc_b16.jsonl > $out/r06g_ab1.log 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms resid_acc=0 resid_acc=1 --rounds 3 --out $out/r06g_ab_resid_acc_fp8_b16.jsonl > $out/r06g_ab2.log 2>&1
python tools/ab_variants.py --precision bf16 --batch 8 --arms resid_acc=0 resid_acc=1 --rounds 3 --out $out/r06g_ab_resid_acc_b8.jsonl > $out/r06g_ab3.log 2>&1
python -m pytest tests/test_golden_gpu.py -x -q -s -k "bf16 or config3 or config4 or config5 or reduced" > $out/r06g_pytest_golden_reduced.txt 2>&1
tail -n 3 $out/r06g_pytest_golden_reduced.txt
cat $out/r06g_ab_resid_acc_*.jsonl | cut -c1-420
