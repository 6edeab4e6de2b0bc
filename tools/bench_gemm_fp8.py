"""A/B of the ResBlock 3x3 convolutions: MXFP8 kernel (k_fp8.hip, tiles 256x320 / 256x256 / 256x128) against the bf16
large-tile kernel (k_gemm_bf16x.hip) on the same shapes, HIP events inside libsdmi (no torch: starts in a second).

    python tools/bench_gemm_fp8.py [--batch 16] [--iters 5]
"""
import argparse
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

# (rows multiplier, cin, h, w, cout): UNet ResBlock convs (rows = 2 * batch: cond + uncond) and VAE ResnetBlock convs (rows = 1 image)
UNET = [(320, 64, 64, 320), (640, 64, 64, 320), (960, 64, 64, 320), (320, 32, 32, 640), (640, 32, 32, 640), (1280, 32, 32, 640),
        (1920, 32, 32, 640), (640, 16, 16, 1280), (1280, 16, 16, 1280), (2560, 16, 16, 1280), (1280, 8, 8, 1280), (2560, 8, 8, 1280)]
VAE = [(512, 64, 64, 512), (512, 128, 128, 512), (512, 256, 256, 256), (256, 256, 256, 256), (256, 512, 512, 128), (128, 512, 512, 128)]
QT = {0: "256x320q", 1: "256x256q", 2: "256x128q"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=2))
    shapes = [(2 * args.batch, c, h, w, n) for c, h, w, n in UNET] + [(1, c, h, w, n) for c, h, w, n in VAE]
    for s in shapes:
        nb, cin, h, w, cout = s
        M, N, K = nb * h * w, cout, cin * 9
        fl = 2.0 * M * N * K
        sd.set_option("fp8_convs", 1)
        row = []
        for cfg, name in QT.items():
            try:
                ms = sd.bench_conv(nb, cin, h, w, cout, 3, 1, 0, cfg, 0, args.iters)
                row.append(f"{name}: {fl / ms / 1e9:6.0f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"{name}: ERR {e}")
        sd.set_option("fp8_convs", 0)
        ms = sd.bench_conv(nb, cin, h, w, cout, 3, 1, 0, -1, 0, args.iters)
        print(f"{str(s):28s} M={M:7d} N={N:5d} K={K:6d} | fp8 " + " | ".join(row) + f" | bf16 auto: {fl / ms / 1e9:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
