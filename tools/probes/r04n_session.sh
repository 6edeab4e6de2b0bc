#!/bin/bash
# round 4, call n: the whole GPU suite + smoke + the default bench line after the attention / bench / default changes
out=gpurun_out/${SDMI_OUT:-r04n}; mkdir -p $out
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $out/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log | cut -c1-300
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
import os
j = json.loads([l for l in open("gpurun_out/" + os.environ.get("SDMI_OUT", "r04n") + "/bench_n1.json") if l.startswith("{")][-1])
print("headline", round(j["value"], 3), "img/s", j["kernel_classes_ms_per_image"], j["launches_profiled_vs_counted"], "frac", round(j["roofline"]["frac"], 3), "cpu", j["cpu_baseline"]["value"] if j["cpu_baseline"] else None)
for s in j["secondary"]:
    print(round(s["value"], 3), s["dtype"], s["config"]["global_batch"], s["config"]["ddim_steps"], s.get("options"), "frac", round(s["roofline"]["frac"], 3), "attn", round(s.get("attention_tflops", 0)), s["kernel_classes_ms_per_image"])
PY
