#!/bin/bash
# round 2, last GPU minutes: the k loops with hand-counted LDS waits (gemm3x_variant=74, gemm_bf16x_variant=3): parity, A/B, PMC.
R=$PWD; out=gpurun_out/r02zz; mkdir -p $out
export SDMI_UNVERIFIED=1
timeout 200 python -m pytest tests -m "gpu and unverified" -q -p no:cacheprovider -n 6 > $out/pytest_unverified.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest_unverified.log
tail -6 $out/pytest_unverified.log
timeout 120 python tools/ab_variants.py --precision fp32 --batch 1 --rounds 3 --out $out/ab_fp32_b1.jsonl \
    --arms gemm3x_variant=2 gemm3x_variant=74 gemm3x_variant=42 > $out/ab_fp32_b1.log 2>&1
echo "ab fp32 rc=$?"; cat $out/ab_fp32_b1.jsonl 2>/dev/null | cut -c1-420
timeout 120 python tools/ab_variants.py --precision bf16 --batch 8 --rounds 2 --out $out/ab_bf16_b8.jsonl \
    --arms gemm_bf16x_variant=0 gemm_bf16x_variant=3 gemm_bf16x_variant=1 > $out/ab_bf16_b8.log 2>&1
echo "ab bf16 rc=$?"; cat $out/ab_bf16_b8.jsonl 2>/dev/null | cut -c1-420
cd /tmp && export TMPDIR=/tmp
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout 100 rocprofv3 --pmc $CNT --output-format csv -d $R/$out/pmc_f32 -- python $R/tools/probes/pmc_variants.py > $R/$out/pmc_f32.log 2>&1
timeout 100 rocprofv3 --pmc $CNT --output-format csv -d $R/$out/pmc_bf16 -- python $R/tools/probes/pmc_variants.py --bf16 > $R/$out/pmc_bf16.log 2>&1
cd $R
python tools/probes/pmc_table.py $out/pmc_f32 > $out/pmc_f32_table.txt 2>&1; python tools/probes/pmc_table.py $out/pmc_bf16 > $out/pmc_bf16_table.txt 2>&1
rm -rf $out/pmc_f32 $out/pmc_bf16
cat $out/pmc_f32.log | grep variant; cat $out/pmc_f32_table.txt; cat $out/pmc_bf16.log | grep variant; cat $out/pmc_bf16_table.txt
