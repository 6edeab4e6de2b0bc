#!/bin/bash
# the driver's bench line, a kernel trace and the HBM-traffic counters of the committed code (no test suite: tools/probes/r02_evidence_session.sh)
set -x
R=$PWD; out=gpurun_out/r02w; mkdir -p $out
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_f32 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof_f32.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 1 $out/pmc_summary.json
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -m gpu -q -p no:cacheprovider -k "conv2d_splitk or cfg1 or geglu" 2>&1 | tail -2
