#!/bin/bash
# round 4, call j: the whole bf16 / fp8 suites + golden bf16 / fp8 model tests with the new attention kernel, then the bf16 B=8 and B=16 bench lines
out=gpurun_out/r04j; mkdir -p $out
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -q -p no:cacheprovider > $out/pytest_bf16_fp8.log 2>&1; echo "bf16+fp8 tests rc=$?"; tail -6 $out/pytest_bf16_fp8.log | cut -c1-300
timeout 1500 python -m pytest tests/test_golden_gpu.py -q -p no:cacheprovider -k "bf16 or fp8 or config3 or config5" > $out/pytest_golden.log 2>&1; echo "golden rc=$?"; tail -6 $out/pytest_golden.log | cut -c1-300
timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out/bench_cfg3.json 2> $out/bench_cfg3.err; echo "bench cfg3 rc=$?"; python - <<'PY'
import json
for f in ("gpurun_out/r04j/bench_cfg3.json",):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, j["value"], j.get("kernel_classes_ms_per_image"), j.get("attention_tflops"), j["roofline"]["achieved"])
    except Exception as e:
        print(f, "unreadable", e)
PY
