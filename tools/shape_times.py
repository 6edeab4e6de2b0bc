#!/usr/bin/env python3
"""Where inside a kernel class the GPU time of one sample_image goes: HIP-event time per launch TAG (class + shape + tile / split-K
choice) of the engine's own launches, inside the model (cold weights, real producers), sorted by share.

    python tools/shape_times.py --config 1 --out gpurun_out/shape_times_fp32_b1.txt
    python tools/shape_times.py --config 2 --ddim-steps 10 --opt gemm_bf16x_variant=1

Engine option profile=2 + dump_profile_tags (csrc/engine.cpp: ProfScope::set_tag).  ms are per IMAGE.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=1)
    ap.add_argument("--ddim-steps", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--out", default="gpurun_out/shape_times.txt")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn

    precision, B, S = {1: ("fp32", 1, 20), 2: ("bf16", 16, 50), 3: ("bf16", 8, 20), 4: ("fp8", 16, 20)}[args.config]
    if args.ddim_steps:
        S = args.ddim_steps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    probe = StableDiffusion(ModelConfig(), device=0)
    flat = probe.pack_weights(syn.SyntheticWeights(cache=False), groups=1)
    probe.close()
    cd = ModelConfig().ctx_dim
    cond = torch.from_numpy(syn.cond_context(0, bench.T_CTX, cd)).to(dev)
    uncond = torch.from_numpy(syn.uncond_context(bench.T_CTX, cd)).to(dev)
    run = bench.Runner(torch, np, dev, 0, precision, B, S, 7.5, cond, uncond, list(range(B)), flat, args.opt, None)
    run.step()
    torch.cuda.synchronize()
    run.sd.set_option("profile_reset", 1)
    run.sd.set_option("profile", 2)
    run.step()
    torch.cuda.synchronize()
    run.sd.set_option("profile", 0)
    tmp = args.out + ".raw"
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    run.sd.set_option("dump_profile_tags", tmp)
    rows = []
    for line in Path(tmp).read_text().splitlines():
        nums, tag = line.split("\t", 1)
        ms, n, fl, by = nums.split()
        rows.append((float(ms), int(n), float(fl), float(by), tag))
    Path(tmp).unlink()
    total = sum(r[0] for r in rows)
    rows.sort(key=lambda r: -r[0])
    out = [f"# config {args.config}: {precision} B={B} S={S} opts={args.opt}; tagged GPU time {total / B:.2f} ms per image; columns: ms/image  share  launches/image  us/launch  TFLOP/s  GB/s(algorithmic)  tag"]
    cum = 0.0
    for ms, n, fl, by, tag in rows:
        cum += ms
        us = ms * 1e3 / max(n, 1)
        out.append(f"{ms / B:9.3f} {100 * ms / total:5.1f}% {100 * cum / total:5.1f}% {n / B:8.1f} {us:9.1f} {fl / (ms * 1e-3) / 1e12 if ms else 0:8.1f} {by / (ms * 1e-3) / 1e9 if ms else 0:8.0f}  {tag}")
    Path(args.out).write_text("\n".join(out) + "\n")
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
