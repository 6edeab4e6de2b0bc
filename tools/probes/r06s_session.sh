#!/bin/bash
# round 6, call s: bf16 attention, QR = 2 with 128-key LDS tiles walked as two 64-key compute chunks (one barrier per 128 keys): tests, isolated, and per image as whole
# libraries (the build of the commit in tools/probes/libs/PREV_COMMIT.txt against this tree, alternating processes)
out=gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "attention" > $out/r06s_pytest_attn.txt 2>&1; tail -n 2 $out/r06s_pytest_attn.txt
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    echo -n "$which $rep: "; python tools/bench_attn.py --bf16 --b16 2>/dev/null | head -1 | cut -c1-110
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06s_bf16_b16_${which}_$rep.jsonl > /dev/null 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06s_bf16_b16_*.jsonl; do echo $f; cut -c1-330 $f; done
