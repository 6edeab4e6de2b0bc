#!/bin/bash
# round 3, GPU call q: the whole GPU suite on the final code (parallel workers: most of its time is the CPU oracle), then the driver's bench line
# and its kernel trace
R=$PWD; out=gpurun_out/r03q; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -n 14 --durations=30 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 $out/pytest_gpu.log | cut -c1-200
timeout 400 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"; cut -c1-600 $out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-200
find $out/prof -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
