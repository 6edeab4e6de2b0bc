#!/usr/bin/env python3
"""A/B of kernel-variant options inside ONE process (one weight load per precision): images/s and the per-class GPU times of
sample_image under each option setting, interleaved (A B C A B C ...) so that clock drift hits every arm alike.

    python tools/ab_variants.py --precision fp32 --batch 1 --arms "gemm3x_variant=2" "gemm3x_variant=10" "gemm3x_variant=42" --rounds 3
    python tools/ab_variants.py --precision bf16 --batch 8 --arms "gemm_bf16x_variant=0" "gemm_bf16x_variant=1"

An arm is a comma-separated list of key=value options; every key of every arm is set for every arm (a key an arm does not name
keeps the value the FIRST arm gives it, so the first arm should be the baseline).  The pseudo-option tunefile=PATH applies every
"M,N,K=cfg,splits" line of a tuning table through the "tune" option (give every arm one, e.g. the built-in table
stable_diffusion_burn_amd/tuning/gfx950_fp32.txt for the baseline).  Prints one JSON line per arm.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp8"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--arms", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=3, help="timed images (batches) per arm, interleaved")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import numpy as np
    import torch
    import bench
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    weights = syn.SyntheticWeights(cache=False)
    probe = StableDiffusion(ModelConfig(), device=0)
    flat = probe.pack_weights(weights, groups=1)
    probe.close()
    ctx_dim = ModelConfig().ctx_dim
    cond = torch.from_numpy(syn.cond_context(0, bench.T_CTX, ctx_dim)).to(dev)
    uncond = torch.from_numpy(syn.uncond_context(bench.T_CTX, ctx_dim)).to(dev)
    run = bench.Runner(torch, np, dev, 0, args.precision, args.batch, args.ddim_steps, 7.5, cond, uncond, list(range(args.batch)), flat, [], None)

    arms = [dict(kv.split("=", 1) for kv in a.split(",")) for a in args.arms]
    base = dict(arms[0])
    for a in arms:
        for k, v in base.items():
            a.setdefault(k, v)

    def apply(a):
        for k, v in a.items():
            if k == "tunefile":
                for line in Path(v).read_text().splitlines():
                    line = line.strip()
                    if "=" in line and not line.startswith("#"):
                        run.sd.set_option("tune", line)
            else:
                run.sd.set_option(k, v)

    times = [[] for _ in arms]
    for i, a in enumerate(arms):   # warm-up: one image per arm (first-launch costs, function attributes)
        apply(a)
        run.step()
        torch.cuda.synchronize()
    for _ in range(args.rounds):
        for i, a in enumerate(arms):
            apply(a)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.step()
            torch.cuda.synchronize()
            times[i].append(time.perf_counter() - t0)
    lines = []
    for i, a in enumerate(arms):
        apply(a)
        _, prof = run.roofline()
        best, med = min(times[i]), sorted(times[i])[len(times[i]) // 2]
        rec = {"arm": args.arms[i], "precision": args.precision, "batch": args.batch, "ddim_steps": args.ddim_steps,
               "img_per_s_best": args.batch / best, "img_per_s_median": args.batch / med, "ms_per_image_median": med / args.batch * 1e3,
               "classes_ms_per_image": {k: round(v["ms"] / args.batch, 3) for k, v in prof.items() if v["ms"] > 0}}
        g = prof.get("conv_gemm_split") if args.precision == "fp32" else prof.get("conv_gemm")
        if g and g["ms"] > 0:
            rec["gemm_tflops"] = g["flops"] / (g["ms"] * 1e-3) / 1e12
        lines.append(json.dumps(rec))
        print(lines[-1], flush=True)
    apply(arms[0])
    run.close()
    if args.out:
        Path(args.out).write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
