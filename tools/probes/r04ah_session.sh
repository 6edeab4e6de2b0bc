#!/bin/bash
# (record of a measured-and-removed lever: the option slab_native and its code are in the history -- commit "fp32 plane GEMM: split-K slabs stored in accumulator order" -- not in the tree)
# round 4, call ah: the whole GPU suite and the headline line on the tree with slab_native (second form: wave-uniform slot decode in the combine kernel)
out=gpurun_out/r04ah; mkdir -p $out
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $out/pytest_gpu.log | tail -3
for o in 1 0; do
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --opt slab_native=$o > $out/bench_n1_slab$o.json 2>/dev/null; python -c "
import sys,json; d=json.load(open('$out/bench_n1_slab$o.json')); k=d['kernel_classes_ms_per_image']; print('slab_native=$o', round(d['value'],4), d['unit'], round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'split', k['conv_gemm_split'], 'reduce', k['splitk_reduce'])"
done | tee $out/headline_ab.txt
