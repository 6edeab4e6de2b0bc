#!/bin/bash
# round 6, call zr: the lean fp32 epilogue (k_gemm_epi.hpp; gemm3x_variant bit 6 = off): GPU tests, interleaved A/B per image at fp32 B = 1 (the headline configuration)
out=gpurun_out
python -m pytest tests -x -q -m gpu > $out/r06zr_pytest_gpu.txt 2>&1; tail -n 3 $out/r06zr_pytest_gpu.txt
python tools/ab_variants.py --precision fp32 --batch 1 --arms gemm3x_variant=66 gemm3x_variant=2 --rounds 4 --out $out/r06zr_ab_fp32_b1.jsonl > $out/r06zr_ab.log 2>&1
cut -c1-420 $out/r06zr_ab_fp32_b1.jsonl
