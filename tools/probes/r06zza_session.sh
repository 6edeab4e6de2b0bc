#!/bin/bash
# round 6, call zza: three bf16 tile-table candidates from r06zz, per image (alternating processes, same box; SDMI_OPTS carries the candidate rows)
out=gpurun_out
for rep in 1 2; do
  python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zza_b16_base_$rep.jsonl > /dev/null 2>&1
  SDMI_OPTS="tune_bf16=8192,1280,5120=100,2" python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zza_b16_rows_$rep.jsonl > /dev/null 2>&1
  python tools/ab_variants.py --precision bf16 --batch 8 --arms cfg_share=1 --rounds 3 --out $out/r06zza_b8_base_$rep.jsonl > /dev/null 2>&1
  SDMI_OPTS="tune_bf16=16384,640,2560=101,1 tune_bf16=16384,640,1920=101,1" python tools/ab_variants.py --precision bf16 --batch 8 --arms cfg_share=1 --rounds 3 --out $out/r06zza_b8_rows_$rep.jsonl > /dev/null 2>&1
done
for f in $out/r06zza_*.jsonl; do echo $f; cut -c1-300 $f; done
