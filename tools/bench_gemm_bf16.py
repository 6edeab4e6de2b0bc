"""A/B the bf16 conv / linear kernels (precision = 1) on model shapes: the 4-wave tiles of k_gemm_bf16.hip
(cfg 0..9) against the 8-wave LDS-DMA tiles of k_gemm_bf16x.hip (cfg 100..103).  HIP events inside libsdmi.

    python tools/bench_gemm_bf16.py [--batch 16] [--quick]
"""
import argparse
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
from tools.autotune import UNET_SHAPES, VAE_SHAPES, mnk  # noqa: E402

XT = {100: "256x320x", 101: "256x256x", 102: "256x128x", 103: "128x320x"}
OLD = {0: "128x128", 7: "128x160", 9: "64x160", 3: "256x128"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (UNet rows are 2x: cond + uncond)")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
    shapes = [(2 * args.batch,) + s[1:] for s in UNET_SHAPES] + [(args.batch,) + s[1:] for s in VAE_SHAPES if s[4] > 4 and s[2] <= 256]
    if args.quick:
        shapes = shapes[:3] + shapes[5:7] + shapes[11:13] + shapes[16:17] + shapes[20:22] + shapes[-3:]
    for s in shapes:
        M, N, K = mnk(s)
        kt = K // 64
        row = []
        best = (1e9, "")
        for cfg, name in list(OLD.items()) + list(XT.items()):
            bm, bn = (int(v) for v in name.rstrip("x").split("x"))
            tiles = -(-M // bm) * -(-N // bn)
            cand = sorted({1} | {sp for sp in (2, 3, 4, 6, 8, 12, 16, 24) if tiles * sp <= 1024 and kt // sp >= 4 and tiles < 256})
            bt = (1e9, 1)
            for sp in cand:
                try:
                    ms = sd.bench_conv(*s, cfg, sp, args.iters)
                except Exception as e:  # noqa: BLE001
                    print("ERR", s, cfg, sp, e)
                    continue
                bt = min(bt, (ms, sp))
            row.append(f"{name}/{bt[1]}: {2.0 * M * N * K / bt[0] / 1e9:6.0f}")
            best = min(best, (bt[0], f"{name}/{bt[1]}"))
        print(f"{str(s):38s} M={M:7d} N={N:5d} K={K:5d} | " + " | ".join(row) + f" | best {best[1]} {2.0 * M * N * K / best[0] / 1e9:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
