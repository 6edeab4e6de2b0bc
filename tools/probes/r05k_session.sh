#!/bin/bash
# round 5, call k: key slices of the fp32 attention kernels (option attn_kv_splits): operator parity, the fp32 model / golden tests with the automatic rule, per-image A/B
out=gpurun_out/r05k; mkdir -p $out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_planes_gpu.py tests/test_clip_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 900 python -m pytest tests/test_golden_gpu.py -x -q -s -k "config2_20_steps or unet_forward_full or per_step or unpadded_contexts_full_size_fp32" 2>&1 | grep -v "^$\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
for b in 1 2; do
  timeout 600 python tools/ab_variants.py --precision fp32 --batch $b --arms "attn_kv_splits=1" "attn_kv_splits=0" --rounds 3 --out $out/ab_fp32_b$b.jsonl > $out/ab_fp32_b$b.log 2>&1; echo "ab fp32 $b rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05k/ab_*.jsonl")):
    for l in open(f):
        r = json.loads(l); c = r["classes_ms_per_image"]
        print(r["precision"], r["batch"], r["arm"], "img/s %.4f (best %.4f)" % (r["img_per_s_median"], r["img_per_s_best"]), {k: c[k] for k in ("conv_gemm_split", "attention", "other") if k in c})
PY
