#!/bin/bash
# round 6, call l: the fp32 latency-chain changes that have no run-time switch (split-K combine, LayerNorm, GroupNorm apply, attention combine, first K / V tile before Q,
# epilogue early loads), measured as whole libraries: the build of commit 8b00d2d (before them) against this tree, two processes each, alternating, same box
out=gpurun_out
new=stable_diffusion_burn_amd/lib/libsdmi.so
cp $new /tmp/libsdmi_new.so
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libs/libsdmi_8b00d2d.so $new; else cp /tmp/libsdmi_new.so $new; fi
    python tools/ab_variants.py --precision fp32 --batch 1 --arms cfg_share=1 --rounds 4 --out $out/r06l_fp32_b1_${which}_$rep.jsonl > $out/r06l_$which$rep.log 2>&1
  done
done
cp /tmp/libsdmi_new.so $new
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "attention or layer_norm or group_norm or model or unet or sample" > $out/r06l_pytest.txt 2>&1
tail -n 3 $out/r06l_pytest.txt
python -m pytest tests/test_golden_gpu.py -x -q -k "config2 or unet_forward or config1 or unpadded_contexts_full_size_fp32" > $out/r06l_pytest_golden_fp32.txt 2>&1
tail -n 2 $out/r06l_pytest_golden_fp32.txt
for f in $out/r06l_fp32_b1_*.jsonl; do echo $f; cut -c1-520 $f; done
