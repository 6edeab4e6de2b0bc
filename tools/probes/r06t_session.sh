#!/bin/bash
# round 6, call t: the final-tree evidence: whole GPU suite, the driver's bench line, then the r06n evidence pass (kernel traces, PMC bytes + matrix-pipe-busy with the tree's digest)
out=gpurun_out
python -m pytest tests -m gpu -x -q > $out/r06t_pytest_gpu.txt 2>&1
grep -n "passed\|failed" $out/r06t_pytest_gpu.txt | tail -2
bash tools/probes/r06n_session.sh $1 > $out/r06t_evidence.log 2>&1
tail -n 4 $out/r06t_evidence.log
cp gpurun_out/r06n/pmc_summary.json profiles/pmc_summary.json     # (bench.py reads profiles/: on this box only; the merged copy travels back in gpurun_out/)
python bench.py > $out/r06t_bench_n1.json 2> $out/r06t_bench_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06t_bench_n1.json'))
r=d['roofline']
print('headline', d['value'], 'frac', r['frac'], 'two_sided', r.get('two_sided',{}).get('frac_two_sided'), 'stale', r.get('traffic_stale'), 'traffic', r.get('traffic'), 'alg', r.get('algorithmic_bytes_per_launch'))
print(d['kernel_classes_ms_per_image']); print(d.get('applied_options'))
for s in d['secondary']: print(s['config']['workload'][:72], round(s['value'],3), round(s['roofline']['frac'],3), s['roofline'].get('two_sided',{}).get('frac_two_sided'), s['roofline'].get('traffic_stale'))
PY
