#!/bin/bash
# round 4, last evidence call: the driver's command on the final tree, the FETCH_SIZE / WRITE_SIZE passes of the same command (profiles/pmc_summary.json -> roofline.traffic),
# and the rocprofv3 kernel traces of the headline and of --config 2 (bf16 batch 16 / 50 steps: shows conv3_gemm_bf16t_kernel beside conv_gemm_bf16x_kernel)
out=gpurun_out/r04zz; mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 1 $out/pmc_summary.json; echo "pmc summary rc=$?"
rm -rf $out/pmc_fetch $out/pmc_write
cp $out/pmc_summary.json profiles/pmc_summary.json     # (on the box only: so that the bench line below reads this round's figure)
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; cut -c1-400 $out/bench_n1.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof.log 2>&1
echo "rocprof fp32 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_fp32_b1.csv; rm -rf $R/$out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof2.log 2>&1
echo "rocprof bf16 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_bf16_b16_s50.csv; rm -rf $R/$out/prof
cd $R
head -6 $out/kernel_stats_fp32_b1.csv | cut -c1-160; head -8 $out/kernel_stats_bf16_b16_s50.csv | cut -c1-160
