#!/usr/bin/env python3
"""bench.py -- images/sec of the MI355X-native SD v1.4 sampling hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `sample_image` call (reference src/model/stablediffusion/mod.rs:51-67)
over one batch of synthetic inputs: 20 DDIM iterations x 2 UNet evaluations (CFG 7.5)
+ VAE decode -> 512x512 RGB u8 ON THE HOST ("to u8 RGB on host"; the D2H copy of
the 786 KB image into pinned memory is inside the timed region), fp32 arithmetic, batch 1 per GPU
(BASELINE.json configs[1]).  Inputs (text embeddings, x_T) are resident in HBM
when the timed region starts; weights are seeded synthetic (no checkpoint / no
network in this environment).

Multi-GPU: one process per GPU; rank 0 owns the prompt embedding and broadcasts
ONE packed buffer [cond(77x768) | uncond(77x768)] over RCCL (backend "nccl");
the image batch is sharded by global image index with no other collective
(SURVEY.md 8e).  Weak scaling: images per GPU fixed.

Prints ONE JSON line on rank 0 (contract in the task statement) including
`roofline` for the dominant kernel (implicit-GEMM conv on fp32 MFMA) measured
live with HIP events on the engine's stream (its `traffic` from the committed
rocprofv3 PMC passes, its `algorithmic_bytes_per_launch` from this run's shapes),
`parity_in_run` (this run's own results against the golden vectors of tests/golden,
after the timed loop), `cpu_baseline` (the fp32 oracle timed on the host cores, rank 0,
N=1 only) and -- at N=1 -- `secondary`: the same measurement for the headline with
every GEMM on the fp32 matrix instruction and for the reduced-precision configurations
BASELINE.json names (the per-GPU shards of configs[4]: MXFP8 convs, batch 16, 20 steps,
and configs[3]: bf16, batch 8, 20 steps; configs[2]: bf16, batch 16, 50 steps), run
after the timed headline loop.
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

FP8_MFMA_PEAK_TFLOPS = 5000.0   # same table, "Peak FP8 MFMA" (dense; the MX-scaled K = 128 instruction)
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA" (dense)
HBM_ACHIEVABLE_TBPS = 6.29      # same guide: "HBM3E peak BW 8.0 TB/s spec; 6.29 TB/s measured (float4 copy)" -- the rate the two-sided roof prices bytes at
SPLIT_PRODUCTS = 6              # k_gemm3x.hip: bf16 MFMAs issued per fp32 16x16x32 block (csrc/k_gemm3x.hip header)
F_UNET = 0.8033e12              # FLOP per UNet forward per sample, T = 77
F_VAE = 2.5145e12               # FLOP per decoded image
T_CTX = 77


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--scale", type=float, default=7.5)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp8"], default="fp32",
                    help="fp32 = BASELINE.json configs[1] (the metric's configuration, default); bf16 = configs[2..3] storage/compute")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="engine option (sdmi_set_option), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bf16 configs[2] / configs[3]-shard measurements after the headline")
    ap.add_argument("--no-parity", action="store_true", help="skip parity_in_run (one more un-timed sample_latent call; the PMC passes count launches per image)")
    ap.add_argument("--pmc-ddim-steps", type=int, default=0, help="PMC passes only: run this many DDIM steps instead of the configuration's (bytes PER LAUNCH do not depend "
                    "on the step count; rocprofv3's counter collection segfaults on the 100 k dispatches of configs[2]'s 50 steps)")
    ap.add_argument("--tune-file", default=str(ROOT / "stable_diffusion_burn_amd" / "tuning" / "gfx950_fp32.txt"))
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=None,
                    help="preset = BASELINE.json configs[i]: 1 fp32 B=1 S=20 (the default headline); 2 bf16 B=16 S=50; 3 bf16 B=8 per GPU S=20 "
                         "(64 images over 8 GPUs); 4 fp8 B=16 per GPU S=20 (128 images over 8 GPUs).  Sets --precision / --batch-per-gpu / --ddim-steps.")
    a = ap.parse_args()
    if a.config is not None:
        a.precision, a.batch_per_gpu, a.ddim_steps = {1: ("fp32", 1, 20), 2: ("bf16", 16, 50), 3: ("bf16", 8, 20), 4: ("fp8", 16, 20)}[a.config]
    if a.pmc_ddim_steps > 0:
        a.ddim_steps, a.no_parity = a.pmc_ddim_steps, True
    return a


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
    sd.forward_diffuser(x, 999, ctx, unc, 7.5)  # untimed warm-up step: first touch converts the cached synthetic weights to torch, the thread pool spins up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sd.forward_diffuser(x, 999, ctx, unc, 7.5)
        ts.append(time.perf_counter() - t0)
    t_step = sorted(ts)[1]
    t0 = time.perf_counter()
    sd.decode_float(x)
    t_vae = time.perf_counter() - t0
    t_img = ddim_steps * t_step + t_vae
    return {"value": 1.0 / t_img, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"median of 3 CFG steps after one warm-up (2 UNet fwd each; {', '.join(f'{t:.2f}' for t in ts)} s) + 1 VAE decode ({t_vae:.2f} s) of the fp32 torch-CPU "
                      f"oracle on {cores} threads, extrapolated to {ddim_steps} steps ({t_img:.1f} s/image)"}


class Runner:
    """One engine + its device-resident inputs; step() = one sample_image over the shard, u8 images on the host."""

    def __init__(self, torch, np, dev, local_rank, precision, B, ddim_steps, scale, cond, uncond, indices, flat, opts, tune_file):
        from stable_diffusion_burn_amd import ModelConfig, StableDiffusion, synthetic as syn
        self.torch, self.B, self.ddim_steps, self.scale = torch, B, ddim_steps, scale
        self.bf16 = precision in ("bf16", "fp8")
        self.fp8 = precision == "fp8"
        cfg = ModelConfig(precision=2 if self.fp8 else 1 if self.bf16 else 0)
        self.cfg = cfg
        self.sd = StableDiffusion(cfg, device=local_rank)
        t0 = time.perf_counter()
        self.sd.load_weights_packed(flat, groups=1)   # the timed path takes embeddings and only decodes: hot-path group only
        self.t_load = time.perf_counter() - t0
        if tune_file and os.path.exists(tune_file) and not self.bf16:
            for line in Path(tune_file).read_text().split():
                if "=" in line and not line.startswith("#"):
                    self.sd.set_option("tune", line.strip())
        for kv in opts:
            k, _, v = kv.partition("=")
            self.sd.set_option(k, v)
        self.sd.set_stream(torch.cuda.current_stream().cuda_stream)   # order the engine's stream behind torch's (no device-wide sync)
        self.context = cond[None].repeat(B, 1, 1).contiguous()            # same prompt for every image
        self.uncond = uncond.contiguous()
        self.latent = torch.from_numpy(np.stack([syn.initial_latent(i, cfg.latent_h, cfg.latent_w) for i in indices])).to(dev)
        self.rgb = torch.empty((B, 8 * cfg.latent_h, 8 * cfg.latent_w, 3), dtype=torch.uint8, device=dev)
        self.rgb_host = torch.empty(self.rgb.shape, dtype=torch.uint8).pin_memory()
        self.np, self.dev, self.indices, self.precision = np, dev, list(indices), precision

    def step(self):
        self.sd.sample_image_dev(self.context.data_ptr(), self.B, T_CTX, self.uncond.data_ptr(), T_CTX, self.scale, self.ddim_steps,
                                 self.latent.data_ptr(), self.rgb.data_ptr())
        self.rgb_host.copy_(self.rgb, non_blocking=True)   # the reference returns Vec<Vec<u8>> on the host (stablediffusion/mod.rs:86-99)

    def timed(self, steps, warmup, barrier):
        for _ in range(warmup):
            self.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        barrier()
        return time.perf_counter() - t0

    def parity_in_run(self):
        """What THIS run produced against the committed golden vectors (tests/golden/*.npz: the oracle's results for exactly these inputs) -- un-timed, after the
        timed loop: the final latent of one more sample_latent call for every image of the shard that has a fixture, and (fp32 headline) the u8 image the timed
        loop left on the host.  The tests assert the bars; this block shows that the numbers printed beside it come from a run that met them."""
        np, torch = self.np, self.torch
        g = ROOT / "tests" / "golden"
        if self.scale != 7.5 or self.cfg.latent_h != 64:
            return None
        key = (self.precision, self.ddim_steps)
        want = {}    # global image index -> (fp64 latent, fixture name[, fp32-oracle latent])
        try:
            if key == ("fp32", 20):
                d = np.load(g / "sd14_synth_cfg2.npz")
                want[0] = (d["latent64"], "sd14_synth_cfg2.npz", d["latents32"][-1], d["rgb_u8"])
            elif key in (("bf16", 50), ("bf16", 20), ("fp8", 20)):
                d = np.load(g / ("sd14_synth_cfg3.npz" if self.ddim_steps == 50 else "sd14_synth_cfg5.npz"))
                for i in (0, 1):
                    want[i] = (d["latent64"][i], "sd14_synth_cfg3.npz" if self.ddim_steps == 50 else "sd14_synth_cfg5.npz")
                more = g / "sd14_synth_more.npz"
                if more.exists():
                    m = np.load(more)
                    for j, i in enumerate(m["index"].tolist()):
                        want[int(i)] = (m["latent64_s50" if self.ddim_steps == 50 else "latent64_s20"][j], "sd14_synth_more.npz")
            else:
                return None
        except Exception as e:  # noqa: BLE001
            return {"error": f"golden fixture unreadable: {e}"}
        pos = [(k, i) for k, i in enumerate(self.indices) if i in want]
        if not pos:
            return None
        lat = torch.empty((self.B, 4, self.cfg.latent_h, self.cfg.latent_w), dtype=torch.float32, device=self.dev)
        self.sd.sample_latent_dev(self.context.data_ptr(), self.B, T_CTX, self.uncond.data_ptr(), T_CTX, self.scale, self.ddim_steps, self.latent.data_ptr(), lat.data_ptr())
        torch.cuda.synchronize()
        lat = lat.cpu().numpy().astype(np.float64)
        out = {"fixtures": sorted({want[i][1] for _, i in pos}), "images": [i for _, i in pos],
               "latent_rel_rms_vs_fp64_oracle": [float(np.sqrt(np.mean((lat[k] - want[i][0]) ** 2) / np.mean(want[i][0] ** 2))) for k, i in pos],
               "latent_max_abs_vs_fp64_oracle": [float(np.abs(lat[k] - want[i][0]).max()) for k, i in pos],
               "bar_asserted_in_tests": {("fp32", 20): "max |latent - fp32 oracle| < 1e-3, u8 image <= 1 LSB (tests/test_golden_gpu.py)",
                                         ("bf16", 50): "rel-RMS <= 1.5 x first measurement (test_config3_bf16_batch16_50_steps)",
                                         ("bf16", 20): "rel-RMS <= 1.5e-2 (test_config4_shard_bf16_batch8_20_steps)",
                                         ("fp8", 20): "rel-RMS <= 6e-2 accuracy budget (test_config5_mxfp8_batch16_20_steps)"}[key]}
        if key == ("fp32", 20):
            k, i = pos[0]
            out["latent_max_abs_vs_fp32_oracle"] = float(np.abs(lat[k] - want[i][2].astype(np.float64)).max())
            du = np.abs(self.rgb_host[k].numpy().astype(np.int16) - want[i][3].astype(np.int16))
            out["u8_image_max_lsb"] = int(du.max())
            out["u8_image_bytes_differing"] = int((du != 0).sum())
            out["u8_image_bytes"] = int(du.size)
        return out

    def roofline(self):
        """Live HIP-event timing of every launch, in a separate un-timed pass (sdmi_profile_stats)."""
        sd = self.sd
        sd.set_option("profile_reset", 1)
        sd.set_option("profile", 2)    # 2: the samples are also accumulated per launch tag (class + shape + tile): the two-sided roof below is per shape
        self.step()
        self.torch.cuda.synchronize()
        sd.set_option("profile", 0)
        prof = sd.profile_stats()
        tags = []
        try:
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                sd.set_option("dump_profile_tags", os.path.join(td, "tags.txt"))
                for line in Path(td, "tags.txt").read_text().splitlines():
                    nums, tag = line.split("\t", 1)
                    ms, n, fl, by = nums.split()
                    tags.append((float(ms), int(n), float(fl), float(by), tag))
        except Exception:  # noqa: BLE001
            tags = []
        # precision = 0: the dominant kernel is the split kernel (fp32 operands as three bf16 terms, six bf16 MFMAs per fp32
        # 16x16x32 block) unless it is switched off (--opt gemm_f32s=0), then the fp32-MFMA kernels
        split = (not self.bf16) and prof["conv_gemm_split"]["ms"] > prof["conv_gemm"]["ms"]
        g = prof["conv_gemm_fp8"] if self.fp8 else prof["conv_gemm_split"] if split else prof["conv_gemm"]
        if g["launches"] <= 0 or g["ms"] <= 0:
            return None, prof
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
        peak = (FP8_MFMA_PEAK_TFLOPS if self.fp8 else BF16_MFMA_PEAK_TFLOPS if self.bf16 else
                BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MFMA_PEAK_TFLOPS)
        kname = ("conv_gemm_fp8x_kernel (implicit-GEMM conv / linear on MXFP8 operands, v_mfma_scale_f32_16x16x128_f8f6f4)" if self.fp8 else
                 "conv_gemm_bf16x_kernel + conv_gemm_bf16_kernel (implicit-GEMM conv/linear, v_mfma_f32_16x16x32_bf16)" if self.bf16 else
                 "conv_gemm3p_kernel / conv_gemm3x_kernel (implicit-GEMM conv/linear in fp32: operands split exactly into 3 bf16 terms -- the weights at "
                 "load, the activations by their producers (planes) or in the k loop --, 6 partial products per multiply on v_mfma_f32_16x16x32_bf16, fp32 accumulation)" if split else
                 "conv_gemm2_kernel + conv_gemm2x_kernel (implicit-GEMM conv/linear, v_mfma_f32_16x16x4_f32)")
        # HBM-side bytes per launch: PMC counters cannot be collected inside this process; `traffic` is the figure of the committed rocprofv3 --pmc passes of this
        # same command (profiles/pmc_summary.json: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units and the gfx950 factor 2 on FETCH_SIZE as
        # MI355X_MICROARCH.md prescribes; Infinity-Cache hits counted), per launch of the dominant kernel class like `achieved`, with the tree it was collected on;
        # `algorithmic_bytes_per_launch` is what this run's launches need by their shapes (source once + weights once + result once, in the stored formats)
        from_profiles = None
        pmc = ROOT / "profiles" / "pmc_summary.json"
        cls = "conv_gemm_fp8" if self.fp8 else "conv_gemm_split" if split else "conv_gemm"
        ckey = f"{self.precision}_b{self.B}_s{self.ddim_steps}"
        if pmc.exists():
            try:
                j = json.loads(pmc.read_text())
                c = j.get("configs", {}).get(ckey) or ({"classes": j["classes"], "commit": j.get("commit")} if ckey == "fp32_b1_s20" and "classes" in j else None)
                t = c["classes"].get(cls, {}).get("hbm_bytes_per_launch") if c else None
                if t:
                    from_profiles = {"hbm_bytes_per_launch": t, "collected_on_commit": c.get("commit"), "kernel_source_digest": c.get("kernel_source_digest"),
                                     # round 6: the matrix pipe's busy share from the same evidence pass (SQ_VALU_MFMA_BUSY_CYCLES), time-weighted over the GEMM / attention kernels
                                     "mfma_busy_pmc": c.get("mfma_busy"), "source": f"profiles/pmc_summary.json configs[{ckey}] classes[{cls}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command"}
            except Exception:  # noqa: BLE001
                from_profiles = None
        roof = {"bound": "mfma", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": from_profiles["hbm_bytes_per_launch"] if from_profiles else None,
                "traffic_unit": "bytes per launch (HBM side incl. Infinity-Cache hits: an upper bound on DRAM bytes); from the committed PMC passes, not this process", "traffic_from_profiles": from_profiles,
                "algorithmic_bytes_per_launch": g["bytes"] / g["launches"] if g.get("bytes") else None,
                "event_pair_overhead_us_subtracted": sd.profile_overhead_us(),
                "achieved_without_event_calibration": g["flops"] / ((g["ms"] + g["launches"] * sd.profile_overhead_us() * 1e-3) * 1e-3) / 1e12,
                "launches_per_image": g["launches"] / self.B, "avg_launch_us": g["ms"] * 1e3 / g["launches"],
                "flop_per_launch": g["flops"] / g["launches"],
                "share_of_gpu_time": g["ms"] / max(1e-9, sum(v["ms"] for v in prof.values()))}
        # The same class against the roof EACH SHAPE can reach (round 6): a launch of 2 M N K flops over its algorithmic bytes is bounded by
        # max(flops / matrix peak, bytes / achievable HBM rate); the short-K Linear / 1x1 layers (K = 320 .. 1280 at N <= 2 K) are HBM-bound by that rule at
        # every batch size, so `frac` above (matrix peak only) is not a bound they can reach.  `frac_two_sided` = sum of per-launch roof times / measured time.
        import re
        sel = [t for t in tags if (t[4].startswith("gemm_fp8 ") if self.fp8 else (t[4].startswith("gemm ") and (self.bf16 or (int(re.search(r"cfg=(\d+)", t[4]).group(1)) >= 200) == split)))]
        if sel and sum(t[0] for t in sel) > 0:
            t_meas = sum(t[0] for t in sel) * 1e-3
            t_mfma = [t[2] / (peak * 1e12) for t in sel]                        # class totals per tag: launches x per-launch time
            t_hbm = [t[3] / (HBM_ACHIEVABLE_TBPS * 1e12) for t in sel]
            t_roof = [max(a, b) for a, b in zip(t_mfma, t_hbm)]
            hb = [i for i in range(len(sel)) if t_hbm[i] > t_mfma[i]]
            roof["two_sided"] = {
                "rule": f"per launch shape: max(2 M N K / {peak:g} TFLOP/s, algorithmic bytes / {HBM_ACHIEVABLE_TBPS} TB/s (measured copy rate, MI355X_MICROARCH.md)); summed over the class",
                "frac_two_sided": sum(t_roof) / t_meas,
                "hbm_bound_shapes": {"share_of_class_time": sum(sel[i][0] for i in hb) * 1e-3 / t_meas, "launches_per_image": sum(sel[i][1] for i in hb) / self.B,
                                     "frac_of_their_hbm_roof": (sum(t_hbm[i] for i in hb) / (sum(sel[i][0] for i in hb) * 1e-3)) if hb else None},
                "mfma_bound_shapes": {"share_of_class_time": 1.0 - sum(sel[i][0] for i in hb) * 1e-3 / t_meas,
                                      "frac_of_mfma_peak": (sum(t_mfma[i] for i in range(len(sel)) if i not in hb) / max(1e-12, sum(sel[i][0] for i in range(len(sel)) if i not in hb) * 1e-3))},
            }
        if from_profiles is not None:
            # the PMC passes cannot run inside this process: say whether the tree they were collected on is this one (digest of csrc/ + the tile tables, build.py)
            try:
                from stable_diffusion_burn_amd import build as _b
                here = _b._digest()[:16]
            except Exception:  # noqa: BLE001
                here = None
            from_profiles["kernel_source_digest_here"] = here
            roof["traffic_stale"] = not (here and from_profiles.get("kernel_source_digest") == here)
        if split:
            o = prof["conv_gemm"]
            roof["peak_note"] = (f"dense bf16 MFMA peak {BF16_MFMA_PEAK_TFLOPS:g} TFLOP/s / {SPLIT_PRODUCTS} matrix instructions per fp32 block; "
                                 f"achieved counts ALGORITHMIC fp32 flops (2 M N K), not the 6x issued bf16 flops")
            roof["clock_note"] = ("the peak is quoted at 2.4 GHz; stamped inside this kernel (option gemm_probe, profiles/r03n_gemm_phase_probe.txt) the shader clock is "
                                  "1.76-1.89 GHz with all 256 CUs on matrix instructions (a power limit), i.e. the bound the chip grants is ~318 TFLOP/s, and the k loop "
                                  "runs at 83 % of it; the rest of a batch-1 launch is prologue, slab store, split-K combine and launch boundaries")
            roof["achieved_vs_fp32_mfma_peak"] = achieved / FP32_MFMA_PEAK_TFLOPS   # > 1 is possible: the fp32 matrix instruction is not used
            roof["issued_bf16_tflops"] = achieved * SPLIT_PRODUCTS
            if o["launches"] > 0 and o["ms"] > 0:
                roof["launches_left_on_fp32_mfma"] = {"launches_per_image": o["launches"] / self.B, "ms_per_image": o["ms"] / self.B,
                                                      "achieved": o["flops"] / (o["ms"] * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS}
        return roof, prof

    def class_summary(self, prof):
        out = {"kernel_classes_ms_per_image": {k: round(v["ms"] / self.B, 3) for k, v in prof.items()},
               "kernel_classes_launches_per_image": {k: v["launches"] / self.B for k, v in prof.items()},
               # every launch of the path is in a class ("other" is measured, not a remainder): launches the profiler saw vs launches the call counted
               "launches_profiled_vs_counted": [sum(v["launches"] for v in prof.values()), self.sd.last_call_stats()["kernels"]]}
        gn = prof["group_norm"]
        if gn["ms"] > 0:
            out["group_norm_algorithmic_GBps"] = gn["bytes"] / (gn["ms"] * 1e-3) / 1e9
        at = prof["attention"]
        if at["ms"] > 0:
            out["attention_tflops"] = at["flops"] / (at["ms"] * 1e-3) / 1e12
        return out

    def close(self):
        self.sd.close()


ARITHMETIC = {
    "fp32": "fp32 storage, fp32 accumulation; conv/linear multiplications: each fp32 operand is the exact sum of three bf16 terms and "
            "the six partial products >= 2^-24 of the product are accumulated in fp32 on the bf16 matrix pipe (csrc/k_gemm3x.hip; per-product "
            "error <= 2^-23 worst case / 2^-28 on average (tests/test_split_oracle_cpu.py), measured against the fp64 oracle: not larger than the fp32 matrix instruction's -- tests/test_ops_gpu.py); "
            "attention, norms and the remaining GEMMs in plain fp32",
    "bf16": "bf16 storage, fp32 accumulation and statistics",
    "fp8": "bf16 storage, fp32 accumulation; on MXFP8 operands (e4m3, E8M0 scale per 32 channels): the ResBlock / ResnetBlock 3x3 convolutions (default; accuracy budget 6e-2 on the "
           "20-step latent) and, with option fp8_linear=1, the UNet's transformer-block Linear layers and 1x1 / up / down convolutions; attention in bf16",
}

def workload_name(precision, B, ddim_steps, scale):
    which = {("fp32", 1, 20): "BASELINE.json configs[1]", ("bf16", 16, 50): "BASELINE.json configs[2]",
             ("bf16", 8, 20): "the per-GPU shard of BASELINE.json configs[3] (64 images over 8 GPUs)",
             ("fp8", 16, 20): "the per-GPU shard of BASELINE.json configs[4] (128 images over 8 GPUs)"}.get((precision, B, ddim_steps), "not a BASELINE.json configuration")
    arith = {"fp32": "fp32", "bf16": "bf16 storage / fp32 accumulate",
             "fp8": "bf16 storage / fp32 accumulate + MXFP8 (e4m3, E8M0 scales per 32 channels) on the ResBlock / ResnetBlock 3x3 convs"}[precision]
    return f"SD v1.4 512x512, {ddim_steps}-step DDIM, CFG={scale}, batch={B} per GPU, {arith} ({which})"


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
    from stable_diffusion_burn_amd import sharding

    B = args.batch_per_gpu
    ctx_dim = ModelConfig().ctx_dim
    weights = syn.SyntheticWeights(cache=(rank == 0 and world == 1 and not args.no_cpu_baseline))
    # the flat fp32 image of the hot-path weights (untimed: numpy RNG, ~3.6 GB); every precision loads the same image.  N > 1: rank 0 generates it
    # once into /dev/shm and the other ranks map it (sharding.share_flat_array) instead of N ranks running the RNG on the same host cores
    t0 = time.perf_counter()

    def make_flat():
        probe = StableDiffusion(ModelConfig(), device=local_rank)
        try:
            return probe.pack_weights(weights, groups=1)
        finally:
            probe.close()

    flat = sharding.share_flat_array(make_flat, rank, world, (lambda: dist.barrier()) if world > 1 else (lambda: None),
                                     f"weights_{os.environ.get('MASTER_PORT', '0')}")
    t_gen = time.perf_counter() - t0

    # ---- inputs: rank 0 owns the prompt embedding; ONE RCCL broadcast ------------------
    packed = torch.empty(((T_CTX + T_CTX) * ctx_dim,), dtype=torch.float32, device=dev)
    if rank == 0:
        p0, _, _ = sharding.pack_prompt(torch.from_numpy(syn.cond_context(0, T_CTX, ctx_dim)),
                                        torch.from_numpy(syn.uncond_context(T_CTX, ctx_dim)))
        packed.copy_(p0)
    sharding.broadcast_prompt(packed, src=0)
    cond, uncond = sharding.unpack_prompt(packed, T_CTX, T_CTX, ctx_dim)
    mine = sharding.shard_range(B * world, rank, world)          # global image indices of this shard

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run = Runner(torch, np, dev, local_rank, args.precision, B, args.ddim_steps, args.scale, cond, uncond, mine, flat, args.opt, args.tune_file)
    elapsed = run.timed(args.steps, args.warmup, barrier)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats = run.sd.last_call_stats()

    roofline, prof = (None, None)
    parity = run.parity_in_run() if rank == 0 and not args.no_parity else None      # (after the timed region; reads the u8 image the last timed step left on the host)
    if rank == 0 and not args.no_roofline:
        roofline, prof = run.roofline()
    bf16 = args.precision in ("bf16", "fp8")
    t_load = run.t_load
    classes = run.class_summary(prof) if prof is not None else {}
    ao = run.sd.applied_options
    applied_options = {"set": [f"{k}={v}" for k, v in ao if k != "tune"], "tune_lines": sum(1 for k, _ in ao if k == "tune"),
                       "from_env_SDMI_OPTS": [f"{k}={v}" for k, v in run.sd.options_from_env]}
    run.close()

    # ---- secondary: the reduced-precision configurations BASELINE.json names, witnessed by the same run ----------
    secondary = []
    if rank == 0 and world == 1 and not args.no_secondary and not bf16 and B == 1 and args.ddim_steps == 20:
        for (prec2, b2, s2, k2, extra) in (("fp32", 1, 20, 5, []), ("fp8", 16, 20, 5, []), ("bf16", 8, 20, 5, []), ("bf16", 16, 50, 5, [])):
            idx = list(range(b2))
            # the headline configuration once more with every GEMM on the fp32 matrix instruction (the split kernel off)
            # (with the tile table that was tuned for those kernels: tuning/gfx950_fp32_mfma.txt)
            opts2, tune2 = (["gemm_f32s=0", "attn_split=0"], str(ROOT / "stable_diffusion_burn_amd" / "tuning" / "gfx950_fp32_mfma.txt")) if prec2 == "fp32" else (extra, None)
            r2 = Runner(torch, np, dev, local_rank, prec2, b2, s2, args.scale, cond, uncond, idx, flat, opts2, tune2)
            e2 = r2.timed(k2, 2, barrier)
            roof2, prof2 = r2.roofline()
            fpi = 2 * s2 * F_UNET + F_VAE
            v2 = k2 * b2 / e2
            entry = {"config": {"workload": workload_name(prec2, b2, s2, args.scale), "global_batch": b2, "ddim_steps": s2,
                                "cfg_scale": args.scale, "context_len": T_CTX},
                     "dtype": "f32" if prec2 == "fp32" else "bf16" if prec2 == "bf16" else "fp8(e4m3, MX)+bf16", "value": v2, "unit": "images/sec", "steps": k2, "warmup": 2, "ms_per_step": e2 / k2 * 1e3,
                     "algorithmic_tflop_per_image": fpi / 1e12, "executed_tflop_per_image": r2.sd.last_call_stats()["flops"] / b2 / 1e12,
                     # round 6: from the flops the engine EXECUTED (the shared CFG prefix and the hoisted projections are not counted); the reference's algorithmic count beside it
                     "whole_path_tflops_per_gpu": v2 * r2.sd.last_call_stats()["flops"] / b2 / 1e12, "whole_path_algorithmic_tflops_per_gpu": v2 * fpi / 1e12,
                     "whole_path_frac_of_bf16_mfma_peak": v2 * r2.sd.last_call_stats()["flops"] / b2 / 1e12 / BF16_MFMA_PEAK_TFLOPS, "roofline": roof2,
                     "applied_options": [f"{k}={v}" for k, v in r2.sd.applied_options],
                     "kernels_per_image": r2.sd.last_call_stats()["kernels"] / b2,
                     "weights_load_s": r2.t_load}
            if prec2 == "fp32":
                entry["config"]["workload"] += "; every GEMM and attention on v_mfma_f32_16x16x4_f32 (options gemm_f32s=0, attn_split=0)"
                entry["whole_path_frac_of_fp32_mfma_peak"] = entry.pop("whole_path_frac_of_bf16_mfma_peak") * BF16_MFMA_PEAK_TFLOPS / FP32_MFMA_PEAK_TFLOPS
            if extra:
                entry["config"]["workload"] += "; option " + ", ".join(extra) + " (MXFP8 also on the transformer blocks' Linear layers and the 1x1 / up / down convs: outside the 6e-2 accuracy budget, 8.1e-2)"
                entry["options"] = extra
            if prec2 != "fp32":
                entry["parity_in_run"] = r2.parity_in_run()
            entry.update(r2.class_summary(prof2))
            secondary.append(entry)
            r2.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(weights, args.ddim_steps)

    if rank == 0:
        images = args.steps * B * world
        value = images / elapsed
        flop_per_image = 2 * args.ddim_steps * F_UNET + F_VAE
        # the bound that applies to the path as it runs: precision 0 multiplies on the bf16 pipe with six matrix instructions per fp32 block
        # (unless the split kernels are switched off), so its matrix bound is 2500 / 6, not the fp32 instruction's 157.3
        split_on = (not bf16) and not any(o.replace(" ", "") in ("gemm_f32s=0",) for o in args.opt)
        peak = BF16_MFMA_PEAK_TFLOPS if bf16 else (BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS if split_on else FP32_MFMA_PEAK_TFLOPS)
        out = {
            "metric": f"images/sec @512x512, {args.ddim_steps}-step DDIM CFG={args.scale:g}, SD v1.4",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8(e4m3, MX)+bf16" if args.precision == "fp8" else "bf16" if bf16 else "f32", "data": "synthetic",
            "arithmetic": ARITHMETIC[args.precision],
            "config": {"workload": workload_name(args.precision, B, args.ddim_steps, args.scale),
                       "global_batch": B * world, "ddim_steps": args.ddim_steps, "cfg_scale": args.scale,
                       "context_len": T_CTX, "parallelism": f"image-sharded x{world}, 1 RCCL broadcast of the text embedding",
                       "cfg": "cond + uncond as one batch-2n forward per step; the layers in front of the first cross attention (identical for the two halves) computed once (cfg_share=1)",
                       "output": "u8 RGB copied to pinned host memory inside the timed region"},
            "algorithmic_tflop_per_image": flop_per_image / 1e12,
            # what the engine's launches actually executed (2 M N K of every GEMM / attention launch): below the reference's count because the part of the UNet in front
            # of the first cross attention is computed once for the two identical halves of a CFG step (option cfg_share) and the text context's K / V
            # projections and the time-embedding MLPs are hoisted out of the step loop
            "executed_tflop_per_image": stats["flops"] / B / 1e12,
            # round 6: priced from the EXECUTED flops (the algorithmic figure, 2.7 % larger, beside it)
            "whole_path_tflops_per_gpu": value / world * stats["flops"] / B / 1e12,
            "whole_path_algorithmic_tflops_per_gpu": value / world * flop_per_image / 1e12,
            "whole_path_frac_of_applicable_mfma_peak": value / world * stats["flops"] / B / 1e12 / peak,
            "applicable_mfma_peak_tflops": peak,
            "applicable_mfma_peak_note": ("dense bf16 MFMA peak" if bf16 else f"dense bf16 MFMA peak / {SPLIT_PRODUCTS} (fp32 operands as three bf16 terms, six partial products)" if split_on
                                          else "fp32 matrix instruction peak"),
            "kernels_per_image": stats["kernels"] / B, "weights_load_s": t_load, "weights_generate_s": t_gen,
            "roofline": roofline, "cpu_baseline": cpu,
            # every engine option this run set beyond the defaults (--opt, the tile table's "tune" lines are counted, SDMI_OPTS from the environment listed by name)
            "applied_options": applied_options,
        }
        out["parity_in_run"] = parity
        out.update(classes)
        if secondary:
            out["secondary"] = secondary
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
