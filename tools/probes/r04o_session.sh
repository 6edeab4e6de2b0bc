#!/bin/bash
# round 4, call o (evidence): the driver's command on this round's defaults, its rocprofv3 kernel trace (fp32 batch 1 and bf16 batch 16 / 50 steps), and the PMC pass over
# representative bf16 launches (the new attention kernel's matrix-pipe occupancy against profiles/r01_pmc_kernels_bf16.txt)
out=gpurun_out/r04o; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; cut -c1-600 $out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof.log 2>&1
echo "rocprof fp32 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_fp32_b1.csv; rm -rf $R/$out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/prof2.log 2>&1
echo "rocprof bf16 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_bf16_b16_s50.csv; rm -rf $R/$out/prof
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$out/pmc_bf16 -- python $R/tools/probes/pmc_shapes.py --bf16 > $R/$out/pmc_shapes_bf16.log 2>&1
echo "pmc rc=$?"
cd $R
{ echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -- python tools/probes/pmc_shapes.py --bf16 (round 4)"; grep -v amdgpu.ids $out/pmc_shapes_bf16.log | grep "^(" ; python tools/pmc_kernels.py $out/pmc_bf16 conv_gemm attn; } > $out/pmc_kernels_bf16.txt 2>&1
rm -rf $out/pmc_bf16
cat $out/pmc_kernels_bf16.txt | cut -c1-220
head -8 $out/kernel_stats_fp32_b1.csv | cut -c1-160; head -8 $out/kernel_stats_bf16_b16_s50.csv | cut -c1-160
