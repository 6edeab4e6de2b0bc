#!/bin/bash
# round 5, call c: the whole GPU suite on the tree with the persistent bf16 tile loop as the default and the explicit q_log2 statement (the three golden tests that
# need sd14_synth_more.npz are deselected: the fixture is still being generated), the driver's bench command (parity_in_run, algorithmic bytes), and the PMC byte
# passes of the headline AND of the reduced-precision configurations (profiles/pmc_summary.json configs[...])
out=gpurun_out/r05c; mkdir -p $out
R=$GRAFT_REPO_ROOT
COMMIT=$1
timeout 1200 python -m pytest tests -m gpu -x -q -k "not config3_bf16 and not config4_shard and not config5_mxfp8" 2>&1 | tail -15 > $out/pytest_gpu_tail.txt; echo "pytest rc=$?"; tail -4 $out/pytest_gpu_tail.txt
timeout 1500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"; cut -c1-300 $out/bench_n1.json; tail -3 $out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
for cfg in 1 2 3 4; do
  key=$(python -c "print({1:'fp32_b1_s20',2:'bf16_b16_s50',3:'bf16_b8_s20',4:'fp8_b16_s20'}[$cfg])")
  imgs=$(python -c "print({1:1,2:16,3:8,4:16}[$cfg])")
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_fetch_$cfg.log 2>&1; echo "pmc fetch cfg $cfg rc=$?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_write_$cfg.log 2>&1; echo "pmc write cfg $cfg rc=$?"
  (cd $R && python tools/pmc_summary.py $out/pmc_fetch_$cfg $out/pmc_write_$cfg $imgs $out/pmc_summary.json $key $COMMIT); echo "pmc summary cfg $cfg rc=$?"
  rm -rf $R/$out/pmc_fetch_$cfg $R/$out/pmc_write_$cfg
done
