#!/bin/bash
# round 6, call zj: per shape, the build of PREV_COMMIT.txt against the tree (alternating processes, same box): which launches the lean epilogue slowed down
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
for which in prev new prev new; do
  if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
  python tools/probes/r06zj_shapes.py 1 >> $out/r06zj_shapes_$which.txt 2>&1
done
cp /tmp/libsdmi_new.so $new
paste -d'|' $out/r06zj_shapes_prev.txt $out/r06zj_shapes_new.txt | cut -c1-200
