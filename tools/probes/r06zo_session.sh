#!/bin/bash
# round 6, call zo: two fragment groups in flight in the lean epilogue -- operator tests (subset), per shape and per image against the build of PREV_COMMIT.txt
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q -k "conv or linear or geglu or persistent or kernel_row" > $out/r06zo_pytest_ops.txt 2>&1; tail -n 2 $out/r06zo_pytest_ops.txt
rm -f $out/r06zo_shapes_*.txt
for which in prev new; do
  if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
  python tools/probes/r06zj_shapes.py 1 >> $out/r06zo_shapes_$which.txt 2>&1
done
paste -d'|' $out/r06zo_shapes_prev.txt $out/r06zo_shapes_new.txt | cut -c1-200
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zo_bf16_b16_${which}_$rep.jsonl > $out/r06zo_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zo_fp8_b16_${which}_$rep.jsonl > $out/r06zo_b$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06zo_*_b*_*.jsonl; do echo $f; cut -c1-330 $f; done
