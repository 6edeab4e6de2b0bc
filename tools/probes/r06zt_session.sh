#!/bin/bash
# round 6, call zt: the time-embedding row of single-sample tiles folded into the per-column accumulator terms (bf16 / MXFP8 kernels) -- operator + golden tests, the new
# lean-vs-general epilogue parity tests, per image against the build of PREV_COMMIT.txt (alternating processes, same box)
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py tests/test_golden_gpu.py tests/test_planes_gpu.py -x -q > $out/r06zt_pytest.txt 2>&1; grep -n "passed\|failed" $out/r06zt_pytest.txt | tail -n 2
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zt_bf16_b16_${which}_$rep.jsonl > $out/r06zt_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zt_fp8_b16_${which}_$rep.jsonl > $out/r06zt_b$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06zt_*_b*_*.jsonl; do echo $f; cut -c1-330 $f; done
