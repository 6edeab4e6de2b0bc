"""Measure implicit-GEMM tile configurations / split-K factors per layer shape on the GPU.

    python tools/autotune.py --quick            # a handful of representative shapes
    python tools/autotune.py --out gpurun_out/tune_fp32.json

Timing is done inside libsdmi with HIP events (sdmi_bench_conv).  The result is a
JSON list of {shape, best cfg, best splits, ms, TF/s, all candidates}; `--emit`
writes the "M,N,K=cfg,splits" lines the engine loads through set_option("tune", ...).
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

TILES = ["128x128", "128x64", "64x64", "256x128", "128x80", "256x80", "64x128", "128x160", "64x80", "64x160"]

# (n, cin, h, w, cout, k, stride, ups): UNet at batch 2 (cond+uncond of one image) + VAE at batch 1
UNET_SHAPES = [
    (2, 320, 64, 64, 320, 3, 1, 0), (2, 640, 64, 64, 320, 3, 1, 0), (2, 960, 64, 64, 320, 3, 1, 0),
    (2, 320, 64, 64, 320, 3, 2, 0), (2, 320, 32, 32, 640, 3, 1, 0), (2, 640, 32, 32, 640, 3, 1, 0),
    (2, 1280, 32, 32, 640, 3, 1, 0), (2, 1920, 32, 32, 640, 3, 1, 0), (2, 960, 32, 32, 640, 3, 1, 0),
    (2, 640, 32, 32, 640, 3, 2, 0), (2, 640, 16, 16, 1280, 3, 1, 0), (2, 1280, 16, 16, 1280, 3, 1, 0),
    (2, 2560, 16, 16, 1280, 3, 1, 0), (2, 1920, 16, 16, 1280, 3, 1, 0), (2, 1280, 16, 16, 1280, 3, 2, 0),
    (2, 1280, 8, 8, 1280, 3, 1, 0), (2, 2560, 8, 8, 1280, 3, 1, 0),
    (2, 1280, 8, 8, 1280, 3, 1, 1), (2, 1280, 16, 16, 1280, 3, 1, 1), (2, 640, 32, 32, 640, 3, 1, 1),
    # 1x1 / linear shapes (as 1x1 convs over the token grid)
    (2, 320, 64, 64, 320, 1, 1, 0), (2, 320, 64, 64, 2560, 1, 1, 0), (2, 1280, 64, 64, 320, 1, 1, 0),
    (2, 640, 32, 32, 640, 1, 1, 0), (2, 640, 32, 32, 5120, 1, 1, 0), (2, 2560, 32, 32, 640, 1, 1, 0),
    (2, 1280, 16, 16, 1280, 1, 1, 0), (2, 1280, 16, 16, 10240, 1, 1, 0), (2, 5120, 16, 16, 1280, 1, 1, 0),
    (2, 1280, 8, 8, 10240, 1, 1, 0), (2, 5120, 8, 8, 1280, 1, 1, 0),
    (2, 640, 64, 64, 320, 1, 1, 0), (2, 960, 64, 64, 320, 1, 1, 0), (2, 1920, 32, 32, 640, 1, 1, 0),
]
VAE_SHAPES = [
    (1, 512, 64, 64, 512, 3, 1, 0), (1, 512, 64, 64, 512, 3, 1, 1), (1, 512, 128, 128, 512, 3, 1, 0),
    (1, 512, 128, 128, 512, 3, 1, 1), (1, 512, 256, 256, 256, 3, 1, 0), (1, 256, 256, 256, 256, 3, 1, 0),
    (1, 256, 256, 256, 256, 3, 1, 1), (1, 256, 512, 512, 128, 3, 1, 0), (1, 128, 512, 512, 128, 3, 1, 0),
    (1, 128, 512, 512, 3, 3, 1, 0), (1, 512, 64, 64, 512, 1, 1, 0),
]
QUICK = [UNET_SHAPES[0], UNET_SHAPES[5], UNET_SHAPES[11], UNET_SHAPES[16], UNET_SHAPES[21], VAE_SHAPES[5], VAE_SHAPES[8]]


def mnk(s):
    n, cin, h, w, cout, k, stride, ups = s
    hin, win = h << ups, w << ups
    pad = 1 if k == 3 else 0
    ho, wo = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
    return n * ho * wo, cout, cin * k * k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--out", default="gpurun_out/tune_fp32.json")
    ap.add_argument("--emit", default="")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--budget-s", type=float, default=240.0)
    ap.add_argument("--shapes-file", default="", help="shapes recorded by the engine (option dump_shapes): 'n,cin,h,w,cout,k,stride,ups count'")
    ap.add_argument("--only", default="", help="n,cin,h,w,cout,k,stride,ups : time just this shape (for rocprofv3 --pmc runs)")
    ap.add_argument("--cfg", type=int, default=-1)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32", help="bf16: k_gemm_bf16.hip tiles 0..9 + k_gemm_bf16x.hip tiles 100..103")
    ap.add_argument("--batch", type=int, default=1, help="images per GPU: scales the n of every shape (the lists are for 1 image)")
    ap.add_argument("--append", action="store_true", help="append to --emit instead of overwriting")
    ap.add_argument("--families", default="old,x,s", help="fp32 candidates: old = k_gemm2.hip tiles 0..9, x = k_gemm2x.hip 100..103, "
                    "s = k_gemm3x.hip 200..205 (fp32 on the bf16 matrix pipe), p = k_gemm3p.hip 300..304 (the same with the activations as planes too)")
    ap.add_argument("--merge", default="", help="existing 'M,N,K=cfg,splits' table: its entry is timed as one more candidate for "
                    "its shape, and --emit writes the whole table with the winners replaced")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (repeatable)")
    args = ap.parse_args()
    bf16 = args.precision == "bf16"
    shapes = QUICK if args.quick else (UNET_SHAPES + (VAE_SHAPES if args.vae else []))
    counts = {}
    if args.shapes_file:
        shapes = []
        for ln in Path(args.shapes_file).read_text().splitlines():
            if ln.strip():
                key, cnt = ln.split()
                sh = tuple(int(v) for v in key.split(","))
                shapes.append(sh)
                counts[sh] = int(cnt)
    if args.batch > 1:
        counts = {(sh[0] * args.batch,) + tuple(sh[1:]): c for sh, c in counts.items()}
        shapes = [(sh[0] * args.batch,) + tuple(sh[1:]) for sh in shapes]
    if bf16:
        shapes = [sh for sh in shapes if sh[1] % 64 == 0]     # Cin = 4 layers run on the fp32 kernel
    sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1) if bf16 else ModelConfig(32, 1, 32, 8, 8, 32))
    sd.set_option("tune_clear", 1)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        sd.set_option(k, v)
    fams = set(args.families.split(","))
    tiles_all = dict(enumerate(TILES)) if (bf16 or "old" in fams) else {}
    if bf16 or "x" in fams:
        tiles_all.update({100: "256x320", 101: "256x256", 102: "256x128", 103: "128x320"})   # k_gemm_bf16x.hip / k_gemm2x.hip
    if not bf16 and "s" in fams:
        tiles_all.update({200: "256x160", 201: "128x320", 202: "256x128", 203: "128x256", 204: "128x160", 205: "128x128"})   # k_gemm3x.hip
    if not bf16 and "p" in fams:
        tiles_all.update({300: "256x160", 301: "256x128", 302: "128x256", 303: "128x160", 304: "128x128", 305: "64x64", 306: "64x128", 307: "64x320", 308: "128x64"})   # k_gemm3p.hip
    merged = {}
    if args.merge:
        for ln in Path(args.merge).read_text().splitlines():
            if "=" in ln and not ln.startswith("#"):
                key, val = ln.strip().split("=")
                merged[key] = tuple(int(v) for v in val.split(","))
    if args.only:
        s = tuple(int(v) for v in args.only.split(","))
        M, N, K = mnk(s)
        ms = sd.bench_conv(*s[:5], k=s[5], stride=s[6], upsample2x=s[7], tile_cfg=args.cfg, splitk=args.splits, iters=args.iters)
        print(f"{s} cfg={args.cfg} splits={args.splits}: {ms:.4f} ms {2.0 * M * N * K / ms / 1e9:.1f} TF")
        return
    results = []
    t_start = time.time()
    for s in shapes:
        M, N, K = mnk(s)
        flops = 2.0 * M * N * K
        kt = (K + 63) // 64 if bf16 else (K + 31) // 32
        cands = []
        for cfg in tiles_all:
            bm, bn = map(int, tiles_all[cfg].split("x"))
            if bf16 and cfg < 100 and M * N > (1 << 24) and cfg in (2, 8):
                continue   # 64-row tiles on very large GEMMs: never competitive, skip the launches
            if cfg >= 100 and not bf16 and s[1] % 32:
                continue   # the large-tile fp32 kernels need Cin % 32 == 0
            tiles = -(-M // bm) * -(-N // bn)
            split_opts = [1]
            for sp in (2, 3, 4, 6, 8, 12, 16, 24, 32):
                if tiles * sp <= 1024 and kt // sp >= 4 and tiles < 400:
                    split_opts.append(sp)
            for sp in split_opts:
                if time.time() - t_start > args.budget_s:
                    break
                try:
                    ms = sd.bench_conv(*s[:5], k=s[5], stride=s[6], upsample2x=s[7], tile_cfg=cfg, splitk=sp,
                                       iters=args.iters if flops < 5e11 else 2)
                except Exception as e:  # noqa: BLE001
                    print(f"  {s} cfg={cfg} sp={sp}: {e}")
                    continue
                cands.append({"cfg": cfg, "tile": tiles_all[cfg] + ("p" if cfg >= 300 else "s" if cfg >= 200 else "x" if cfg >= 100 else ""), "splits": sp, "ms": ms, "tflops": flops / ms / 1e9})
        prev = merged.get(f"{M},{N},{K}")
        if prev:
            try:
                ms = sd.bench_conv(*s[:5], k=s[5], stride=s[6], upsample2x=s[7], tile_cfg=prev[0], splitk=prev[1], iters=args.iters if flops < 5e11 else 2)
                cands.append({"cfg": prev[0], "tile": f"table:{prev[0]}", "splits": prev[1], "ms": ms, "tflops": flops / ms / 1e9})
            except Exception as e:  # noqa: BLE001
                print(f"  {s} table entry {prev}: {e}")
        if not cands:
            continue
        best = min(cands, key=lambda c: c["ms"])
        auto_ms = sd.bench_conv(*s[:5], k=s[5], stride=s[6], upsample2x=s[7], tile_cfg=-1, splitk=0, iters=args.iters)
        results.append({"shape": s, "count": counts.get(s, 1), "M": M, "N": N, "K": K, "best": best, "heuristic_ms": auto_ms,
                        "heuristic_tflops": flops / auto_ms / 1e9, "cands": cands})
        print(f"{str(s):44s} M={M:6d} N={N:5d} K={K:5d}  best {best['tile']:8s} x{best['splits']:<2d} "
              f"{best['ms']:8.3f} ms {best['tflops']:6.1f} TF | heuristic {auto_ms:8.3f} ms "
              f"{flops / auto_ms / 1e9:6.1f} TF", flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(results, indent=1))
    if args.emit:
        for r in results:
            merged[f"{r['M']},{r['N']},{r['K']}"] = (r["best"]["cfg"], r["best"]["splits"])
        with open(args.emit, "a" if args.append else "w") as f:
            if args.merge:
                for key, (cfg, sp) in merged.items():
                    f.write(f"{key}={cfg},{sp}\n")
            else:
                for r in results:
                    f.write(f"{r['M']},{r['N']},{r['K']}={r['best']['cfg']},{r['best']['splits']}\n")
    tot = sum(2.0 * r["M"] * r["N"] * r["K"] * r["count"] for r in results)
    tb = sum(r["best"]["ms"] * r["count"] for r in results)
    th = sum(r["heuristic_ms"] * r["count"] for r in results)
    print(f"launch-weighted: best {tb:.1f} ms, heuristic/current {th:.1f} ms")
    print(f"sum over shapes: best {tot / tb / 1e9:.1f} TF/s, heuristic {tot / th / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
