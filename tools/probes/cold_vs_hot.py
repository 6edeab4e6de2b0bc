#!/usr/bin/env python3
"""Per-shape time of the split GEMM families with the operands cache-hot (back-to-back launches of one layer: what tools/autotune.py measured
so far) and with the WEIGHTS evicted between launches (option bench_cold: what a layer sees inside the model, where 5 GB of weight planes stream
through the 256 MB Infinity Cache between two uses).  For each shape: the k_gemm3x.hip table entry and the k_gemm3p.hip table entry."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
import autotune  # noqa: E402


def table(path):
    t = {}
    for ln in Path(path).read_text().split():
        if "=" in ln and not ln.startswith("#"):
            k, v = ln.split("=")
            t[k] = tuple(int(x) for x in v.split(","))
    return t


def main():
    t3x = table(ROOT / "stable_diffusion_burn_amd" / "tuning" / "gfx950_fp32.txt")
    tp = table(sys.argv[1]) if len(sys.argv) > 1 else {}
    shapes = [(2, 320, 64, 64, 320, 3, 1, 0), (2, 640, 32, 32, 640, 3, 1, 0), (2, 1280, 16, 16, 1280, 3, 1, 0), (2, 2560, 16, 16, 1280, 3, 1, 0),
              (2, 1280, 8, 8, 1280, 3, 1, 0), (2, 2560, 8, 8, 1280, 3, 1, 0), (1, 320, 1, 8192, 2560, 1, 1, 0), (1, 1280, 1, 8192, 320, 1, 1, 0),
              (1, 1280, 1, 512, 10240, 1, 1, 0), (1, 5120, 1, 512, 1280, 1, 1, 0), (1, 256, 256, 256, 256, 3, 1, 0)]
    sd = StableDiffusion(ModelConfig(32, 1, 32, 8, 8, 32))
    sd.set_option("tune_clear", 1)
    print(f"{'shape':40s} {'M,N,K':22s} {'kernel':10s} {'hot us':>8s} {'cold us':>8s} {'cold/hot':>8s} {'cold TF':>8s}")
    for s in shapes:
        M, N, K = autotune.mnk(s)
        key = f"{M},{N},{K}"
        flops = 2.0 * M * N * K
        for name, tab in (("3x", t3x), ("3p", tp)):
            if key not in tab:
                continue
            cfg, sp = tab[key]
            res = {}
            for cold in (0, 1):
                sd.set_option("bench_cold", cold)
                res[cold] = sd.bench_conv(*s[:5], k=s[5], stride=s[6], upsample2x=s[7], tile_cfg=cfg, splitk=sp, iters=6)
            sd.set_option("bench_cold", 0)
            extra = f"  prefetched {res[2] * 1e3:8.1f} us" if 2 in res else ""
            print(f"{str(s):40s} {key:22s} {name + ':' + str(cfg) + 'x' + str(sp):10s} {res[0] * 1e3:8.1f} {res[1] * 1e3:8.1f} {res[1] / res[0]:8.2f} {flops / res[1] / 1e9:8.1f}{extra}")
    sd.close()


if __name__ == "__main__":
    main()
