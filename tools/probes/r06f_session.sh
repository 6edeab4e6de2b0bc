#!/bin/bash
# round 6, call f: bf16 attention forms per image (attn_bf16_variant 0 = round 5's form, 7 = the automatic choice of the round-6 forms), interleaved in one process,
# then the attention operator tests and the reduced-precision golden fixtures on the new default
out=gpurun_out
python tools/ab_variants.py --precision bf16 --batch 16 --arms attn_bf16_variant=0 attn_bf16_variant=7 attn_bf16_variant=2 --rounds 3 --out $out/r06f_ab_attn_bf16_variant_b16.jsonl > $out/r06f_ab1.log 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms attn_bf16_variant=0 attn_bf16_variant=7 --rounds 3 --out $out/r06f_ab_attn_bf16_variant_fp8_b16.jsonl > $out/r06f_ab2.log 2>&1
python tools/ab_variants.py --precision bf16 --batch 8 --arms attn_bf16_variant=0 attn_bf16_variant=7 --rounds 3 --out $out/r06f_ab_attn_bf16_variant_b8.jsonl > $out/r06f_ab3.log 2>&1
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06f_pytest_bf16_fp8.txt 2>&1
python -m pytest tests/test_golden_gpu.py -x -q -s -k "bf16 or config3 or config4 or config5 or reduced" > $out/r06f_pytest_golden_reduced.txt 2>&1
tail -3 $out/r06f_pytest_bf16_fp8.txt $out/r06f_pytest_golden_reduced.txt
cat $out/r06f_ab_attn_bf16_variant_*.jsonl | cut -c1-400
