#!/bin/bash
# round 4, call p: GroupNorm statistics from the split-K combine (fp32) -- model tests, golden fp32 tests, A/B of the headline bench
out=gpurun_out/r04p; mkdir -p $out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_planes_gpu.py -q -p no:cacheprovider -x -s > $out/pytest_model.log 2>&1; echo "model tests rc=$?"; grep "kernels per forward" $out/pytest_model.log | cut -c1-200; tail -3 $out/pytest_model.log | cut -c1-300
timeout 1200 python -m pytest tests/test_golden_gpu.py -q -p no:cacheprovider -x -k "config1 or config2 or unet_forward_full or unpadded_contexts_full_size_fp32 or per_step" -s > $out/pytest_golden.log 2>&1; echo "golden rc=$?"; grep "final latent\|Tc=77" $out/pytest_golden.log | cut -c1-200; tail -2 $out/pytest_golden.log | cut -c1-300
for v in 0 1 0 1; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --opt gn_from_reduce=$v 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('gn_from_reduce=$v', round(j['value'], 4), 'img/s', j['kernels_per_image'], {k: j['kernel_classes_ms_per_image'][k] for k in ('group_norm', 'splitk_reduce', 'conv_gemm_split')})"; done
