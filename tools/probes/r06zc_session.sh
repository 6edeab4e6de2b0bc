#!/bin/bash
# round 6, call zc: the reduced-precision configurations as their own bench lines on the final tree (python bench.py --config N), and the per-shape tables of the final tree
out=gpurun_out
for c in 2 3 4; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $out/r06zc_bench_config${c}_final_tree.json 2> $out/r06zc_bench_config$c.err; done
python tools/shape_times.py --config 2 --ddim-steps 10 --out $out/r06zc_shape_times_bf16_b16.txt > /dev/null 2>&1
python tools/shape_times.py --config 4 --out $out/r06zc_shape_times_fp8_b16.txt > /dev/null 2>&1
python tools/shape_times.py --config 1 --out $out/r06zc_shape_times_fp32_b1.txt > /dev/null 2>&1
python - <<'PY'
import json
for c in (2,3,4):
    d=json.load(open(f'gpurun_out/r06zc_bench_config{c}_final_tree.json'))
    r=d['roofline']
    print(c, round(d['value'],3), round(r['achieved']), round(r['frac'],3), r.get('two_sided',{}).get('frac_two_sided'), d.get('attention_tflops'), d.get('group_norm_algorithmic_GBps'), (d.get('parity_in_run') or {}).get('latent_rel_rms_vs_fp64_oracle'))
PY
head -12 $out/r06zc_shape_times_bf16_b16.txt | cut -c1-150
