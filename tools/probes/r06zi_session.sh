#!/bin/bash
# round 6, call zi: the lean epilogue (interior tiles, bf16 output: every lane-derived address once per tile), linear-layer addressing without divisions, the next tile's bias requested
# in front of the epilogue's stores -- operator tests, per-image A/B against the build of PREV_COMMIT.txt (alternating processes, same box), short-K table
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06zi_pytest_ops.txt 2>&1; tail -n 2 $out/r06zi_pytest_ops.txt
python tools/probes/r04ab_shortk.py > $out/r06zi_shortk.txt 2>&1; cut -c1-330 $out/r06zi_shortk.txt
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zi_bf16_b16_${which}_$rep.jsonl > $out/r06zi_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zi_fp8_b16_${which}_$rep.jsonl > $out/r06zi_b$which$rep.log 2>&1
    python tools/ab_variants.py --precision bf16 --batch 1 --arms cfg_share=1 --rounds 3 --out $out/r06zi_bf16_b1_${which}_$rep.jsonl > $out/r06zi_c$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06zi_*_b*_*.jsonl; do echo $f; cut -c1-300 $f; done
