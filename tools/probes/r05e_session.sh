#!/bin/bash
# round 5, call e: the PMC byte passes of the reduced-precision configurations that segfaulted inside rocprofv3 in call c (100 k dispatches): fewer DDIM steps
# (bytes per launch do not depend on the step count), no parity call; and the headline once more without the parity call (launches per image were doubled by it)
out=gpurun_out/r05e; mkdir -p $out
R=$GRAFT_REPO_ROOT
COMMIT=$1
cp $R/gpurun_out/r05c/pmc_summary.json $R/$out/pmc_summary.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for cfg in 1 2 3 4; do
  key=$(python -c "print({1:'fp32_b1_s20',2:'bf16_b16_s50',3:'bf16_b8_s20',4:'fp8_b16_s20'}[$cfg])")
  imgs=$(python -c "print({1:1,2:16,3:8,4:16}[$cfg])")
  extra=$(python -c "print({1:'--no-parity',2:'--pmc-ddim-steps 4',3:'--pmc-ddim-steps 5',4:'--pmc-ddim-steps 5'}[$cfg])")
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_fetch_$cfg.log 2>&1; echo "pmc fetch cfg $cfg rc=$?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $extra > $R/$out/pmc_write_$cfg.log 2>&1; echo "pmc write cfg $cfg rc=$?"
  (cd $R && python tools/pmc_summary.py $out/pmc_fetch_$cfg $out/pmc_write_$cfg $imgs $out/pmc_summary.json $key $COMMIT); echo "pmc summary cfg $cfg rc=$?"
  rm -rf $R/$out/pmc_fetch_$cfg $R/$out/pmc_write_$cfg
done
