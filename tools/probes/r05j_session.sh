#!/bin/bash
# round 5, call j: GEGLU gate in the plane GEMM's epilogue (wave-column pairs): operator parity over every plane tile, the fp32 model tests, per-image A/B at batch 1 / 4
out=gpurun_out/r05j; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "geglu" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_planes_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_golden_gpu.py -x -q -s -k "config2_20_steps or unet_forward_full or per_step" 2>&1 | grep -v "^$" | tail -8
for b in 1 4; do
  timeout 600 python tools/ab_variants.py --precision fp32 --batch $b --arms "geglu_fuse=0" "geglu_fuse=1" --rounds 3 --out $out/ab_fp32_b$b.jsonl > $out/ab_fp32_b$b.log 2>&1; echo "ab fp32 $b rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05j/ab_*.jsonl")):
    for l in open(f):
        r = json.loads(l); c = r["classes_ms_per_image"]
        print(r["precision"], r["batch"], r["arm"], "img/s %.4f (best %.4f)" % (r["img_per_s_median"], r["img_per_s_best"]), {k: c[k] for k in ("conv_gemm_split", "geglu", "attention", "group_norm", "splitk_reduce") if k in c})
PY
