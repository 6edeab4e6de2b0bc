#!/bin/bash
# round 6, call k: bf16 attention with K / V requested two tiles ahead (attn_bf16_variant bit 3), isolated and per image; operator tests of the form
out=gpurun_out
for v in 7 15 7 15; do echo -n "variant $v: "; python tools/bench_attn.py --bf16 --b16 --bf16-variants=$v 2>/dev/null | head -1 | sed 's/.*| variant/variant/'; done > $out/r06k_attn_bf16_deep_prefetch.txt 2>&1
SDMI_OPTS="attn_bf16_variant=15" python -m pytest tests/test_bf16_gpu.py -x -q -k "attention" > $out/r06k_pytest_attn_variant15.txt 2>&1
tail -n 2 $out/r06k_pytest_attn_variant15.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms attn_bf16_variant=7 attn_bf16_variant=15 --rounds 3 --out $out/r06k_ab_attn_deep_b16.jsonl > $out/r06k_ab1.log 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms attn_bf16_variant=7 attn_bf16_variant=15 --rounds 3 --out $out/r06k_ab_attn_deep_fp8_b16.jsonl > $out/r06k_ab2.log 2>&1
cat $out/r06k_attn_bf16_deep_prefetch.txt; cat $out/r06k_ab_attn_deep_*.jsonl | cut -c1-330
