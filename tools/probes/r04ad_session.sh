#!/bin/bash
# (record only: the temporary switches / variant this call measured are not in the tree; see profiles/README.md, "Round 4")
# round 4, call ad: what the fixed cost of a short-K bf16 GEMM tile is made of -- timing-only ablations of the epilogue (SDMI_EPI_ABL: 1 no epilogue at all, 2 no global stores)
out=gpurun_out/r04ad; mkdir -p $out
for a in 0 2 1; do echo "SDMI_EPI_ABL=$a"; SDMI_EPI_ABL=$a timeout 200 python tools/probes/r04ab_shortk.py 2>/dev/null | grep "tile 100"; done | tee $out/shortk_epilogue_ablation.txt
