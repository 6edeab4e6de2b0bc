#!/bin/bash
# round 6, call za: the staggered DMA issue inside the persistent tile loop of the 256 x 256 / 256 x 128 tiles (19 - 27 SGPRs spilled to lanes): whole libraries, alternating
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06za_bf16_b16_${which}_$rep.jsonl > /dev/null 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06za_fp8_b16_${which}_$rep.jsonl > /dev/null 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06za_*.jsonl; do echo $f; cut -c1-300 $f; done
