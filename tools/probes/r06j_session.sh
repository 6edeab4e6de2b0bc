#!/bin/bash
# round 6, call j: after the explicit LDS-DMA wait in front of every k-loop barrier: determinism of the batch-32 forward, the whole GPU suite, the driver's bench line
out=gpurun_out
python tools/probes/r06i_debug.py > $out/r06j_debug.txt 2>&1
python -m pytest tests -m gpu -x -q > $out/r06j_pytest_gpu.txt 2>&1
tail -n 4 $out/r06j_pytest_gpu.txt
python bench.py > $out/r06j_bench_n1.json 2> $out/r06j_bench_n1.err
cat $out/r06j_debug.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06j_bench_n1.json'))
print(d['value'], d['roofline']['frac'], d['kernel_classes_ms_per_image'])
for s in d['secondary']: print(s['config']['workload'][:70], s['value'], s['roofline']['frac'], s['roofline'].get('two_sided',{}).get('frac_two_sided'))
PY
