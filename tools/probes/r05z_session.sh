#!/bin/bash
# round 5, evidence call on the final tree: the whole GPU suite, the driver's bench command, the PMC byte passes of all four configurations
# (profiles/pmc_summary.json -> roofline.traffic), and the rocprofv3 kernel traces of the headline and of --config 2 / 4
out=gpurun_out/${OUTDIR:-r05z}; mkdir -p $out
R=$GRAFT_REPO_ROOT
COMMIT=$1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > $out/pytest_gpu_tail.txt; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $out/pytest_gpu_tail.txt
cd /tmp && export TMPDIR=/tmp
for cfg in 1 2 3 4; do
  key=$(python -c "print({1:'fp32_b1_s20',2:'bf16_b16_s50',3:'bf16_b8_s20',4:'fp8_b16_s20'}[$cfg])")
  imgs=$(python -c "print({1:1,2:16,3:8,4:16}[$cfg])")
  extra=$(python -c "print({1:'--no-parity',2:'--pmc-ddim-steps 4',3:'--pmc-ddim-steps 5',4:'--pmc-ddim-steps 5'}[$cfg])")
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_fetch_$cfg.log 2>&1; echo "pmc fetch cfg $cfg rc=$?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_write_$cfg.log 2>&1; echo "pmc write cfg $cfg rc=$?"
  (cd $R && python tools/pmc_summary.py $out/pmc_fetch_$cfg $out/pmc_write_$cfg $imgs $out/pmc_summary.json $key $COMMIT > $out/pmc_summary_$cfg.txt); echo "pmc summary cfg $cfg rc=$?"
  rm -rf $R/$out/pmc_fetch_$cfg $R/$out/pmc_write_$cfg
done
cd $R
cp $out/pmc_summary.json profiles/pmc_summary.json     # (on the box only: so that the bench line below reads this call's figures)
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench rc=$?"; cut -c1-300 $out/bench_n1.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity > $R/$out/prof.log 2>&1
echo "rocprof fp32 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_fp32_b1.csv; rm -rf $R/$out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity > $R/$out/prof2.log 2>&1
echo "rocprof bf16 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_bf16_b16_s50.csv; rm -rf $R/$out/prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity > $R/$out/prof4.log 2>&1
echo "rocprof fp8 rc=$?"
f=$(find $R/$out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_fp8_b16_s20.csv; rm -rf $R/$out/prof
cd $R
head -6 $out/kernel_stats_fp32_b1.csv | cut -c1-160; head -8 $out/kernel_stats_bf16_b16_s50.csv | cut -c1-160
