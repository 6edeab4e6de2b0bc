#!/bin/bash
# (record only: the temporary switches / variant this call measured are not in the tree; see profiles/README.md, "Round 4")
# round 4, call z: what the bf16 attention's tile time is made of -- timing-only ablations of attn_bf16_kernel<40, 8> (SDMI_ATTN_ABL bits: 1 no exp / cvt, 2 no maxima,
# 4 no global loads / LDS stores, 8 no barrier, 16 no V fragment reads, 32 no K fragment reads); results of ablated runs are wrong by construction
out=gpurun_out/r04z; mkdir -p $out
for a in 0 1 3 4 12 16 48 60 15 63; do
  echo "ABL=$a: $(SDMI_ATTN_ABL=$a timeout 120 python tools/bench_attn.py --bf16 --b16 2>/dev/null | grep -v amdgpu.ids | head -2 | cut -c1-120 | tr '\n' ' ')"
done | tee $out/attn_ablation.txt
