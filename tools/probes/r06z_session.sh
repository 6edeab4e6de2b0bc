#!/bin/bash
# round 6, call z: staggered DMA issue, which launches: gemm_bf16x_variant 1 (off) / 5 (kernel-row convolutions + one-tile forms) / 13 (also: launches of >= 8 k tiles leave the
# tile loop for the staggered one-tile form), per image, interleaved
out=gpurun_out
SDMI_OPTS="gemm_bf16x_variant=13" python -m pytest tests/test_bf16_gpu.py -x -q -k "conv or linear or kernel_row or persistent or geglu" > $out/r06z_pytest_variant13.txt 2>&1; tail -n 2 $out/r06z_pytest_variant13.txt
python tools/ab_variants.py --precision bf16 --batch 16 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 gemm_bf16x_variant=13 --rounds 3 --out $out/r06z_ab_stagger_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision fp8 --batch 16 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 gemm_bf16x_variant=13 --rounds 3 --out $out/r06z_ab_stagger_fp8_b16.jsonl > /dev/null 2>&1
python tools/ab_variants.py --precision bf16 --batch 8 --arms gemm_bf16x_variant=1 gemm_bf16x_variant=5 gemm_bf16x_variant=13 --rounds 3 --out $out/r06z_ab_stagger_b8.jsonl > /dev/null 2>&1
cat $out/r06z_ab_stagger_*.jsonl | cut -c1-330
