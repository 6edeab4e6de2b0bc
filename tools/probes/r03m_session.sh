#!/bin/bash
# (record of what was run: the last command's option xcd_map was removed afterwards -- its result is profiles/r03m_ab_fp32_b1_xcd_map.jsonl)
# round 3, GPU call m: the numbers the precision = 1 / 2 bars are set from (outputs saved, the fp64 side is evaluated off the box:
# tools/probes/r03m_compare.py), the operator-level tests of the fp8_linear kernels, and the HBM-traffic counters of the committed code
R=$PWD; out=gpurun_out/r03m; mkdir -p $out
timeout 400 python tools/probes/r03m_dump.py > $out/dump.log 2>&1; echo "dump rc=$?"; grep -v amdgpu.ids $out/dump.log | tail -8
timeout 200 python -m pytest tests/test_fp8_gpu.py -q -p no:cacheprovider -s -k "grid_operands or quantising or conv3x3_mxfp8_quantises" > $out/pytest_fp8_ops.log 2>&1
echo "pytest fp8 ops rc=$?"; grep -E "passed|failed|identical|Error|error" $out/pytest_fp8_ops.log | cut -c1-300 | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary > $R/$out/pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write 1 $out/pmc_summary.json
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
timeout 200 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 2 --out $out/ab_fp32_b1_xcd_map.jsonl --arms xcd_map=0 xcd_map=1 > $out/ab_xcd.log 2>&1
echo "ab rc=$?"; cut -c1-500 $out/ab_fp32_b1_xcd_map.jsonl
