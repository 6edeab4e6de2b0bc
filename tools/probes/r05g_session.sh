#!/bin/bash
# round 5, call g: GroupNorm apply passes with their first rows requested in front of gn_finalize: parity (group_norm tests of the three precisions) and per-image
# class times against the previous build's numbers (r05f: bf16 B=16 3.62 ms, fp32 B=1 19.2 ms group_norm per image)
out=gpurun_out/r05g; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_fp8_gpu.py tests/test_planes_gpu.py -x -q -k "group_norm or plane_producers" 2>&1 | tail -3
for pb in "bf16 16" "fp8 16" "bf16 8" "fp32 1"; do set -- $pb
  timeout 600 python tools/ab_variants.py --precision $1 --batch $2 --arms "gn_unroll=2" --rounds 3 --out $out/ab_$1_b$2.jsonl > $out/ab_$1_b$2.log 2>&1; echo "ab $1 $2 rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05g/ab_*.jsonl")):
    for l in open(f):
        r = json.loads(l); print(r["precision"], r["batch"], "img/s %.4f" % r["img_per_s_median"], {k: v for k, v in r["classes_ms_per_image"].items()})
PY
