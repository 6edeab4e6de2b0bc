#!/bin/bash
# round 5, call d: launch geometry of the bf16 / MXFP8 GroupNorm passes (chunks per sample bounded by target_wgs / n, workgroup size, loads in flight per thread):
# per-image class times under each setting, interleaved in one process (tools/ab_variants.py), and the parity tests of the operator under the new default candidates
out=gpurun_out/r05d; mkdir -p $out
L="gn_target_wgs=0,gn_max_threads=1024,gn_unroll=1"
timeout 900 python tools/ab_variants.py --precision bf16 --batch 16 --arms "$L" "gn_target_wgs=512,gn_max_threads=1024,gn_unroll=1" "gn_target_wgs=512,gn_max_threads=1024,gn_unroll=2" "gn_target_wgs=512,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=4" "gn_target_wgs=2048,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=256,gn_unroll=4" --rounds 2 --out $out/ab_gn_bf16_b16.jsonl > $out/ab_gn_bf16_b16.log 2>&1; echo "ab bf16 b16 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r05d/ab_gn_bf16_b16.jsonl",):
    for l in open(f):
        r = json.loads(l); print(r["arm"], "img/s %.3f" % r["img_per_s_median"], "group_norm ms/img %.3f" % r["classes_ms_per_image"]["group_norm"])
PY
timeout 900 python tools/ab_variants.py --precision bf16 --batch 8 --arms "$L" "gn_target_wgs=512,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=4" --rounds 2 --out $out/ab_gn_bf16_b8.jsonl > $out/ab_gn_bf16_b8.log 2>&1; echo "ab bf16 b8 rc=$?"
timeout 900 python tools/ab_variants.py --precision fp8 --batch 16 --arms "$L" "gn_target_wgs=512,gn_max_threads=1024,gn_unroll=2" "gn_target_wgs=512,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=2" "gn_target_wgs=1024,gn_max_threads=512,gn_unroll=4" --rounds 2 --out $out/ab_gn_fp8_b16.jsonl > $out/ab_gn_fp8_b16.log 2>&1; echo "ab fp8 b16 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r05d/ab_gn_bf16_b8.jsonl", "gpurun_out/r05d/ab_gn_fp8_b16.jsonl"):
    for l in open(f):
        r = json.loads(l); print(r["precision"], r["batch"], r["arm"], "img/s %.3f" % r["img_per_s_median"], "group_norm ms/img %.3f" % r["classes_ms_per_image"]["group_norm"])
PY
for t in "gn_target_wgs=512 gn_max_threads=512 gn_unroll=2" "gn_target_wgs=1024 gn_max_threads=512 gn_unroll=4"; do
  SDMI_OPTS="$t" timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q -k "group_norm" 2>&1 | tail -2
done
# fp32 batch 1 (the headline): more workgroups per GroupNorm pass than the 64 KB cut gives at n = 2 samples
timeout 900 python tools/ab_variants.py --precision fp32 --batch 1 --arms "gn32_min_wgs=0" "gn32_min_wgs=256" "gn32_min_wgs=512" "gn32_min_wgs=1024" --rounds 3 --out gpurun_out/r05d/ab_gn_fp32_b1.jsonl > gpurun_out/r05d/ab_gn_fp32_b1.log 2>&1; echo "ab fp32 b1 rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r05d/ab_gn_fp32_b1.jsonl"):
    r = json.loads(l); print(r["arm"], "img/s %.4f" % r["img_per_s_median"], "group_norm ms/img %.3f" % r["classes_ms_per_image"]["group_norm"])
PY
SDMI_OPTS="gn32_min_wgs=512" timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "group_norm" 2>&1 | tail -2
