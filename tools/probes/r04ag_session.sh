#!/bin/bash
# (record of a measured-and-removed lever: the option slab_native and its code are in the history -- commit "fp32 plane GEMM: split-K slabs stored in accumulator order" -- not in the tree)
# round 4, call ag: split-K slabs of the plane GEMM in accumulator order (slab_native): parity of the fp32 suites, then the headline with the option on / off
out=gpurun_out/r04ag; mkdir -p $out
timeout 600 python -m pytest tests/test_planes_gpu.py -q -p no:cacheprovider -x > $out/pytest_planes.log 2>&1; echo "planes tests rc=$?"; tail -2 $out/pytest_planes.log | cut -c1-200
for o in 1 0 1 0; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --opt slab_native=$o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes_ms_per_image']; print('slab_native=$o', round(d['value'],4), d['unit'], round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'split', k['conv_gemm_split'], 'reduce', k['splitk_reduce'])"
done | tee $out/headline_ab.txt
timeout 600 python -m pytest tests/test_golden_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -q -p no:cacheprovider -x -k "not bf16 and not fp8 and not reduced and not mxfp8" > $out/pytest_fp32.log 2>&1; echo "fp32 tests rc=$?"; tail -2 $out/pytest_fp32.log | cut -c1-200
