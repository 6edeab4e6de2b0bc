#!/bin/bash
# round 5, call f: GroupNorm geometry, second sweep around the first one's winners (defaults now 512 / 512 / 2 and gn32_min_wgs = 256)
out=gpurun_out/r05f; mkdir -p $out
timeout 900 python tools/ab_variants.py --precision bf16 --batch 16 --arms "gn_target_wgs=512,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=256,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=256,gn_max_threads=1024,gn_unroll=2" "gn_target_wgs=384,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=256,gn_max_threads=512,gn_unroll=4" "gn_target_wgs=768,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=0,gn_max_threads=1024,gn_unroll=1" --rounds 2 --out $out/ab_gn_bf16_b16.jsonl > $out/ab_gn_bf16_b16.log 2>&1; echo "ab bf16 b16 rc=$?"
timeout 900 python tools/ab_variants.py --precision fp32 --batch 1 --arms "gn32_min_wgs=256" "gn32_min_wgs=128" "gn32_min_wgs=192" "gn32_min_wgs=320" "gn32_min_wgs=384" "gn32_min_wgs=0" --rounds 3 --out $out/ab_gn_fp32_b1.jsonl > $out/ab_gn_fp32_b1.log 2>&1; echo "ab fp32 b1 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r05f/ab_gn_bf16_b16.jsonl", "gpurun_out/r05f/ab_gn_fp32_b1.jsonl"):
    for l in open(f):
        r = json.loads(l); print(r["precision"], r["batch"], r["arm"], "img/s %.4f" % r["img_per_s_median"], "group_norm ms/img %.3f" % r["classes_ms_per_image"]["group_norm"])
PY
