#!/usr/bin/env python3
"""bench.py -- images/sec of the MI355X-native SD v1.4 sampling hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `sample_image` call (reference src/model/stablediffusion/mod.rs:51-67)
over one batch of synthetic inputs: 20 DDIM iterations x 2 UNet evaluations (CFG 7.5)
+ VAE decode -> 512x512 RGB u8, fp32 arithmetic, batch 1 per GPU
(BASELINE.json configs[1]).  Inputs (text embeddings, x_T) are resident in HBM
when the timed region starts; weights are seeded synthetic (no checkpoint / no
network in this environment).

Multi-GPU: one process per GPU; rank 0 owns the prompt embedding and broadcasts
ONE packed buffer [cond(77x768) | uncond(77x768)] over RCCL (backend "nccl");
the image batch is sharded by global image index with no other collective
(SURVEY.md 8e).  Weak scaling: images per GPU fixed.

Prints ONE JSON line on rank 0 (contract in the task statement) including
`roofline` for the dominant kernel (implicit-GEMM conv on fp32 MFMA) measured
live with HIP events on the engine's stream, and `cpu_baseline` (the fp32 oracle
timed on the host cores, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA" (dense)
F_UNET = 0.8033e12              # FLOP per UNet forward per sample, T = 77
F_VAE = 2.5145e12               # FLOP per decoded image


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--scale", type=float, default=7.5)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE.json configs[1] (the metric's configuration, default); bf16 = configs[2..3] storage/compute")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="engine option (sdmi_set_option), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tune-file", default=str(ROOT / "stable_diffusion_burn_amd" / "tuning" / "gfx950_fp32.txt"))
    return ap.parse_args()


def cpu_baseline(weights, ddim_steps: int) -> dict:
    """Oracle ("port" of the reference's arithmetic, oracle/sd_oracle.py) on the host cores.

    Bounded sample: ONE CFG step (2 UNet forwards, batch 1, T = Tu = 77) + ONE VAE decode
    of the full-size fp32 model; the per-image time is extrapolated as
    ddim_steps * t_step + t_vae (the 20 steps are identical in cost).
    """
    import torch
    from oracle.sd_oracle import Dims, StableDiffusionOracle
    from stable_diffusion_burn_amd import synthetic as syn

    # torch-CPU conv/GEMM stops scaling (and collapses) far below 256 threads: cap the pool and
    # report the cores actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = StableDiffusionOracle(weights, syn.alphas_cumprod(), Dims(), torch.float32)
    x = torch.from_numpy(syn.initial_latent(0))[None]
    ctx = torch.from_numpy(syn.cond_context(0))[None]
    unc = torch.from_numpy(syn.uncond_context())
    sd.unet.forward(x, 999, ctx)  # untimed: first touch converts the cached synthetic weights to torch
    t0 = time.perf_counter()
    sd.forward_diffuser(x, 999, ctx, unc, 7.5)
    t_step = time.perf_counter() - t0
    t0 = time.perf_counter()
    sd.decode_float(x)
    t_vae = time.perf_counter() - t0
    t_img = ddim_steps * t_step + t_vae
    return {"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 CFG step (2 UNet fwd, {t_step:.2f} s) + 1 VAE decode ({t_vae:.2f} s) of the fp32 torch-CPU "
                      f"oracle on {cores} threads, extrapolated to {ddim_steps} steps ({t_img:.1f} s/image)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import numpy as np
    import torch  # BEFORE libsdmi: the loader then shares torch's HIP runtime (same SONAME)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn

    B = args.batch_per_gpu
    T = Tu = 77
    bf16 = args.precision == "bf16"
    cfg = ModelConfig(precision=1 if bf16 else 0)
    sd = StableDiffusion(cfg, device=local_rank)
    weights = syn.SyntheticWeights(cache=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    t0 = time.perf_counter()
    sd.load_weights(weights, clip=False, vae_encoder=False)   # the timed path takes embeddings and only decodes
    t_load = time.perf_counter() - t0
    if os.path.exists(args.tune_file) and not bf16:
        for line in Path(args.tune_file).read_text().split():
            if "=" in line and not line.startswith("#"):
                sd.set_option("tune", line.strip())
    for kv in args.opt:
        k, _, v = kv.partition("=")
        sd.set_option(k, v)

    # ---- inputs: rank 0 owns the prompt embedding; ONE RCCL broadcast ------------------
    from stable_diffusion_burn_amd import sharding
    packed = torch.empty(((T + Tu) * cfg.ctx_dim,), dtype=torch.float32, device=dev)
    if rank == 0:
        p0, _, _ = sharding.pack_prompt(torch.from_numpy(syn.cond_context(0, T, cfg.ctx_dim)),
                                        torch.from_numpy(syn.uncond_context(Tu, cfg.ctx_dim)))
        packed.copy_(p0)
    sharding.broadcast_prompt(packed, src=0)
    cond, uncond = sharding.unpack_prompt(packed, T, Tu, cfg.ctx_dim)
    context = cond[None].repeat(B, 1, 1).contiguous()            # same prompt for every image
    uncond = uncond.contiguous()
    mine = sharding.shard_range(B * world, rank, world)          # global image indices of this shard
    latent = torch.from_numpy(np.stack([syn.initial_latent(i, cfg.latent_h, cfg.latent_w) for i in mine])).to(dev)
    rgb = torch.empty((B, 8 * cfg.latent_h, 8 * cfg.latent_w, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        sd.sample_image_dev(context.data_ptr(), B, T, uncond.data_ptr(), Tu, args.scale, args.ddim_steps,
                            latent.data_ptr(), rgb.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats = sd.last_call_stats()

    # ---- roofline of the dominant kernel: live HIP-event timing, separate (untimed) pass ----
    roofline = None
    prof = None
    if rank == 0 and not args.no_roofline:
        sd.set_option("profile_reset", 1)
        sd.set_option("profile", 1)
        step()
        sd.set_option("profile", 0)
        prof = sd.profile_stats()
        g = prof["conv_gemm"]
        if g["launches"] > 0 and g["ms"] > 0:
            achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
            traffic = None
            pmc = ROOT / "profiles" / "pmc_summary.json"
            if pmc.exists():
                try:
                    traffic = json.loads(pmc.read_text()).get("conv_gemm_hbm_bytes_per_launch")
                except Exception:  # noqa: BLE001
                    traffic = None
            peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
            kname = ("conv_gemm_bf16_kernel (implicit-GEMM conv/linear, v_mfma_f32_16x16x32_bf16)" if bf16 else
                     "conv_gemm2_kernel + conv_gemm2x_kernel (implicit-GEMM conv/linear, v_mfma_f32_16x16x4_f32)")
            if bf16:
                traffic = None  # the committed PMC summary is for the fp32 kernel
            roofline = {"bound": "mfma", "kernel": kname,
                        "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": traffic,
                        "launches_per_image": g["launches"] / B,
                        "avg_launch_us": g["ms"] * 1e3 / g["launches"],
                        "flop_per_launch": g["flops"] / g["launches"],
                        "share_of_gpu_time": g["ms"] / max(1e-9, sum(v["ms"] for v in prof.values()))}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(weights, args.ddim_steps)

    if rank == 0:
        images = args.steps * B * world
        value = images / elapsed
        flop_per_image = 2 * args.ddim_steps * F_UNET + F_VAE
        out = {
            "metric": "images/sec @512x512, 20-step DDIM CFG=7.5, SD v1.4",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": (f"SD v1.4 512x512, {args.ddim_steps}-step DDIM, CFG={args.scale}, batch={B} per GPU, "
                                    + ("bf16 storage / fp32 accumulate (BASELINE.json configs[2..3] family; NOT the headline fp32 configuration)"
                                       if bf16 else "fp32 (BASELINE.json configs[1])")),
                       "global_batch": B * world, "ddim_steps": args.ddim_steps, "cfg_scale": args.scale,
                       "context_len": T, "parallelism": f"image-sharded x{world}, 1 RCCL broadcast of the text embedding"},
            "algorithmic_tflop_per_image": flop_per_image / 1e12,
            "whole_path_tflops_per_gpu": value / world * flop_per_image / 1e12,
            "whole_path_frac_of_mfma_peak": value / world * flop_per_image / 1e12 / (BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS),
            "kernels_per_image": stats["kernels"] / B, "weights_load_s": t_load,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if prof is not None:
            out["kernel_classes_ms_per_image"] = {k: round(v["ms"] / B, 3) for k, v in prof.items()}
            gn = prof["group_norm"]
            if gn["ms"] > 0:
                out["group_norm_algorithmic_GBps"] = gn["bytes"] / (gn["ms"] * 1e-3) / 1e9
            at = prof["attention"]
            if at["ms"] > 0:
                out["attention_tflops"] = at["flops"] / (at["ms"] * 1e-3) / 1e12
        print(json.dumps(out), flush=True)

    sd.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
