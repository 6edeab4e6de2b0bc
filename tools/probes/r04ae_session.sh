#!/bin/bash
# round 4, call ae: the bf16 large-tile epilogue with v_cvt_pk_bf16_f32 and the LDS round trip of fragment group mi + 1 behind the conversion / stores of group mi
out=gpurun_out/r04ae; mkdir -p $out
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x > $out/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -2 $out/pytest_bf16.log | cut -c1-200
timeout 200 python tools/probes/r04ab_shortk.py 2>/dev/null | grep "tile 10[01]" | tee $out/shortk.txt
for i in 1 2; do
  timeout 300 python bench.py --precision bf16 --batch-per-gpu 16 --ddim-steps 20 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 b16 s20', d['value'], d['unit'], d['ms_per_step'])"
done
