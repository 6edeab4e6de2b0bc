#!/bin/bash
# round 6, call zv: the tile loop's EARLY form (next tile's first k tile DMA'd during the last k iteration, epilogue on its own scratch, no barrier between k loop and epilogue;
# gemm_bf16x_variant bit 4 = off) -- operator tests, per image against the build of PREV_COMMIT.txt (alternating processes, same box) and by the run-time switch
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06zv_pytest_ops.txt 2>&1; grep -n "passed\|failed" $out/r06zv_pytest_ops.txt | tail -n 2
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zv_bf16_b16_${which}_$rep.jsonl > $out/r06zv_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zv_fp8_b16_${which}_$rep.jsonl > $out/r06zv_b$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
python tools/ab_variants.py --precision bf16 --batch 16 --arms gemm_bf16x_variant=21 gemm_bf16x_variant=5 --rounds 4 --out $out/r06zv_ab_switch_bf16_b16.jsonl > $out/r06zv_sw.log 2>&1
for f in $out/r06zv_*_b*_*.jsonl $out/r06zv_ab_switch_bf16_b16.jsonl; do echo $f; cut -c1-330 $f; done
