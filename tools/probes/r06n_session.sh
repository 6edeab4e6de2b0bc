#!/bin/bash
# round 6, call n: the evidence pass on the tree named by $1 (commit): rocprofv3 kernel traces of the bench command per configuration, the PMC byte passes
# (FETCH_SIZE / WRITE_SIZE, separate passes) summarised into pmc_summary.json WITH the kernel-source digest bench.py compares against, and a matrix-pipe-busy pass
# (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) for the dominant kernels of each configuration
out=gpurun_out/r06n; mkdir -p $out
R=$GRAFT_REPO_ROOT
COMMIT=$1
cp $R/profiles/pmc_summary.json $R/$out/pmc_summary.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for cfg in 1 2 3 4; do
  key=$(python -c "print({1:'fp32_b1_s20',2:'bf16_b16_s50',3:'bf16_b8_s20',4:'fp8_b16_s20'}[$cfg])")
  imgs=$(python -c "print({1:1,2:16,3:8,4:16}[$cfg])")
  extra=$(python -c "print({1:'--no-parity',2:'--pmc-ddim-steps 4',3:'--pmc-ddim-steps 5',4:'--pmc-ddim-steps 5'}[$cfg])")
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace_$cfg -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-parity > $R/$out/trace_$cfg.log 2>&1; echo "trace cfg $cfg rc=$?"
  cp $(find $R/$out/trace_$cfg -name "*kernel_stats.csv" | head -1) $R/$out/r06n_bench_${key}_kernel_stats.csv 2>/dev/null
  rm -rf $R/$out/trace_$cfg
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_fetch_$cfg.log 2>&1; echo "pmc fetch cfg $cfg rc=$?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_write_$cfg.log 2>&1; echo "pmc write cfg $cfg rc=$?"
  (cd $R && python tools/pmc_summary.py $out/pmc_fetch_$cfg $out/pmc_write_$cfg $imgs $out/pmc_summary.json $key $COMMIT) > $R/$out/r06n_pmc_summary_$key.txt 2>&1; echo "pmc summary cfg $cfg rc=$?"
  rm -rf $R/$out/pmc_fetch_$cfg $R/$out/pmc_write_$cfg
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$out/pmc_mfma_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary --no-parity $(python -c "print({1:'--pmc-ddim-steps 4',2:'--pmc-ddim-steps 2',3:'--pmc-ddim-steps 2',4:'--pmc-ddim-steps 2'}[$cfg])") > $R/$out/pmc_mfma_$cfg.log 2>&1; echo "pmc mfma cfg $cfg rc=$?"
  (cd $R && python tools/pmc_kernels.py $out/pmc_mfma_$cfg conv_gemm conv3_gemm attn --json=$out/pmc_summary.json:$key) > $R/$out/r06n_mfma_busy_$key.txt 2>&1
  rm -rf $R/$out/pmc_mfma_$cfg
done
cd $R; cat $out/r06n_pmc_summary_*.txt | head -60; head -12 $out/r06n_mfma_busy_*.txt
