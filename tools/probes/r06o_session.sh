#!/bin/bash
# round 6, call o: the GEGLU gate with packed fp32 arithmetic (no run-time switch): the build of commit 6e2bef9 against this tree, alternating processes, same box
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q -k "geglu" > $out/r06o_pytest_geglu.txt 2>&1; tail -n 2 $out/r06o_pytest_geglu.txt
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_6e2bef9.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06o_bf16_b16_${which}_$rep.jsonl > $out/r06o_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06o_fp8_b16_${which}_$rep.jsonl > $out/r06o_b$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06o_*_b16_*.jsonl; do echo $f; cut -c1-420 $f; done
