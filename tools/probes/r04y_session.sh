#!/bin/bash
# (record only: the temporary switches / variant this call measured are not in the tree; see profiles/README.md, "Round 4")
# round 4, call y: bf16 attention with the softmax interleaved into the matrix instructions of the same wave (exponentials of step st + 1 beside P V of step st, row maxima beside K Q^T)
out=gpurun_out/r04y; mkdir -p $out
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -p no:cacheprovider -x -k "attention" > $out/pytest_attn.log 2>&1; echo "attn tests rc=$?"; tail -4 $out/pytest_attn.log | cut -c1-300
timeout 300 python tools/bench_attn.py --bf16 --b16 > $out/bench_attn_b16.txt 2>&1; grep -v amdgpu.ids $out/bench_attn_b16.txt | cut -c1-200
