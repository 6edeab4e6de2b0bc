#!/bin/bash
# round 6, call zl: the lean epilogue + Linear addressing + bias ahead of the stores, second build (the one-call accumulator init restored for the one-tile kernels): operator
# tests, per shape and per image against the build of PREV_COMMIT.txt (alternating processes, same box)
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06zl_pytest_ops.txt 2>&1; tail -n 2 $out/r06zl_pytest_ops.txt
rm -f $out/r06zl_shapes_*.txt
for which in prev new; do
  if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
  python tools/probes/r06zj_shapes.py 1 >> $out/r06zl_shapes_$which.txt 2>&1
done
paste -d'|' $out/r06zl_shapes_prev.txt $out/r06zl_shapes_new.txt | cut -c1-200
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zl_bf16_b16_${which}_$rep.jsonl > $out/r06zl_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zl_fp8_b16_${which}_$rep.jsonl > $out/r06zl_b$which$rep.log 2>&1
    python tools/ab_variants.py --precision bf16 --batch 1 --arms cfg_share=1 --rounds 3 --out $out/r06zl_bf16_b1_${which}_$rep.jsonl > $out/r06zl_c$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06zl_*_b*_*.jsonl; do echo $f; cut -c1-330 $f; done
