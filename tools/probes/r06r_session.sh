#!/bin/bash
out=gpurun_out
python tools/probes/r06r_fp8_sweep.py > $out/r06r_fp8_sweep.txt 2>&1
python -m pytest tests/test_fp8_gpu.py -x -q > $out/r06r_pytest_fp8.txt 2>&1; tail -n 2 $out/r06r_pytest_fp8.txt
python -m pytest tests/test_golden_gpu.py -x -q -k "config5 or precision2 or reduced" > $out/r06r_pytest_golden_p2.txt 2>&1; tail -n 2 $out/r06r_pytest_golden_p2.txt
# the bf16 table entries of r06p, per image: the three shapes forced back to the cost model's choice through SDMI_OPTS against the table, alternating processes
for rep in 1 2; do
  SDMI_OPTS="tune_bf16=8192,1280,11520=103,1 tune_bf16=2048,1280,2560=9,1" python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06r_bf16_b16_model_choice_$rep.jsonl > /dev/null 2>&1
  python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06r_bf16_b16_table_$rep.jsonl > /dev/null 2>&1
done
python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06r_fp8_b16.jsonl > /dev/null 2>&1
cat $out/r06r_fp8_sweep.txt; for f in $out/r06r_*.jsonl; do echo $f; cut -c1-300 $f; done
