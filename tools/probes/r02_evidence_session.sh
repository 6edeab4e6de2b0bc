#!/bin/bash
# round-2 evidence run: full GPU suite, the driver's bench line, split-K A/B, kernel trace, HBM-traffic counters
set -x
R=$PWD; out=gpurun_out/r02t; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout 200 python bench.py --no-secondary --no-cpu-baseline --opt gemm3x_variant=6 > $out/bench_2stage.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_f32 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof_f32.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 1 $out/pmc_summary.json
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
python -c "
import json
j=json.load(open('$out/bench_default.json')); print(j['value'], j['kernel_classes_ms_per_image']); print(j['roofline']); print(j['cpu_baseline'])
for s in j.get('secondary', []): print(s['config']['workload'][:90], s['value'])
for f in ():
    b=json.load(open('$out/bench_fused%d.json' % f)); print('splitk_fused', f, b['value'])
"
du -sh $out
