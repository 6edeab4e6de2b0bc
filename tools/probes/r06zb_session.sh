#!/bin/bash
# round 6, call zb: bf16 attention, the second wave of every SIMD one phase late (attn_bf16_variant bit 3): tests, isolated (separate processes), per image
out=gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "attention" > $out/r06zb_pytest_attn.txt 2>&1; tail -n 2 $out/r06zb_pytest_attn.txt
for v in 7 15 7 15; do echo -n "variant $v: "; python tools/bench_attn.py --bf16 --b16 --bf16-variants=$v 2>/dev/null | head -1 | sed 's/.*| variant/variant/'; done > $out/r06zb_attn_lag.txt 2>&1
cat $out/r06zb_attn_lag.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms attn_bf16_variant=7 attn_bf16_variant=15 --rounds 3 --out $out/r06zb_ab_attn_lag_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms attn_bf16_variant=7 attn_bf16_variant=15 --rounds 3 --out $out/r06zb_ab_attn_lag_fp8_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision bf16 --batch 8 --arms attn_bf16_variant=7 attn_bf16_variant=15 --rounds 3 --out $out/r06zb_ab_attn_lag_b8.jsonl > /dev/null 2>&1
cat $out/r06zb_ab_attn_lag_*.jsonl | cut -c1-330
