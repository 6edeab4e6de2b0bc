#!/bin/bash
# round 3, GPU call k: the driver's command on this round's defaults, its rocprofv3 kernel trace, and the model-level / full-size parity tests
out=gpurun_out/r03k; mkdir -p $out
timeout 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; cut -c1-900 $out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-200
rm -rf $out/prof
timeout 700 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -q -p no:cacheprovider -n 4 -x > $out/pytest_model_golden.log 2>&1
echo "pytest rc=$?"; tail -5 $out/pytest_model_golden.log | cut -c1-300
