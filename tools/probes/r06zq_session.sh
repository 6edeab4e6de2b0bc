#!/bin/bash
# round 6, call zq: the residual tile of interior tiles loaded in straight-line batches (k_gemm_bf16_epi.hpp: gemm_acc_resid_bf16) -- operator tests, per image against the
# build of PREV_COMMIT.txt (alternating processes, same box), per-shape table inside the model
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
python -m pytest tests/test_bf16_gpu.py tests/test_fp8_gpu.py -x -q > $out/r06zq_pytest_ops.txt 2>&1; tail -n 2 $out/r06zq_pytest_ops.txt
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_prev.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision bf16 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zq_bf16_b16_${which}_$rep.jsonl > $out/r06zq_a$which$rep.log 2>&1
    python tools/ab_variants.py --precision fp8 --batch 16 --arms cfg_share=1 --rounds 3 --out $out/r06zq_fp8_b16_${which}_$rep.jsonl > $out/r06zq_b$which$rep.log 2>&1
    python tools/ab_variants.py --precision bf16 --batch 1 --arms cfg_share=1 --rounds 3 --out $out/r06zq_bf16_b1_${which}_$rep.jsonl > $out/r06zq_c$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
for f in $out/r06zq_*_b*_*.jsonl; do echo $f; cut -c1-330 $f; done
python tools/shape_times.py --config 2 --ddim-steps 10 --out $out/r06zq_shape_times_bf16_b16.txt > /dev/null 2>&1; grep resid $out/r06zq_shape_times_bf16_b16.txt | head -20 | cut -c1-150
