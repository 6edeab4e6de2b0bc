#!/bin/bash
# round 4, call af: the whole GPU suite, the driver's command and the bf16 / precision-2 kernel traces on the tree with the 2-byte-scratch epilogue (k_gemm_bf16_epi.hpp, shared by
# k_gemm_bf16x.hip, k_gemm_bf16t.hip and now k_fp8.hip) and v_cvt_pk_bf16_f32 in every bf16-producing kernel
out=gpurun_out/r04af; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $out/pytest_gpu.log | tail -3
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; python -c "
import json; d=json.load(open('$out/bench_n1.json')); print(d['value'], d['roofline']['frac']); [print(s.get('value'), s.get('ms_per_step')) for s in d['secondary']]"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof2.log 2>&1
echo "rocprof bf16 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_bf16_b16_s50.csv; rm -rf $R/$out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof4.log 2>&1
echo "rocprof fp8 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_fp8_b16_s20.csv; rm -rf $R/$out/prof
