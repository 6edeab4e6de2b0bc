#!/bin/bash
# round 5, call h: the shared CFG prefix (option cfg_share): parity (shared vs two full forwards on the tiny model in the three precisions, the full-size golden tests
# of the fp32 headline with the default) and the per-image A/B in the four configurations
out=gpurun_out/r05h; mkdir -p $out
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "shared_cfg or sample_latent or sample_image" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_golden_gpu.py -x -q -s -k "config2_20_steps or unpadded or per_step_drift or bf16_full_size" 2>&1 | grep -v "^$" | tail -12
for pb in "fp32 1" "bf16 16" "bf16 8" "fp8 16"; do set -- $pb
  timeout 600 python tools/ab_variants.py --precision $1 --batch $2 --arms "cfg_share=0" "cfg_share=1" --rounds 3 --out $out/ab_$1_b$2.jsonl > $out/ab_$1_b$2.log 2>&1; echo "ab $1 $2 rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05h/ab_*.jsonl")):
    for l in open(f):
        r = json.loads(l); c = r["classes_ms_per_image"]
        print(r["precision"], r["batch"], r["arm"], "img/s %.4f (best %.4f)" % (r["img_per_s_median"], r["img_per_s_best"]), {k: c[k] for k in ("conv_gemm", "conv_gemm_split", "attention", "group_norm", "layer_norm", "other") if k in c})
PY
