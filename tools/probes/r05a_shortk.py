"""Round 5: the r04ab sweep (time against K at M = 131 072 = the 64x64 level at CFG batch 32, fixed N and tile) under each gemm_bf16x_variant:
0 = round 4's kernels, 1 = persistent tile loop, 2 = epilogue without the LDS transpose, 3 = both.  Also N = 1280 as the fused GEGLU projection would see it is not
reachable through bench_conv; the GEGLU form is timed end to end by the A/B of the session script."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
for variant in (0, 1, 2, 3):
    sd.set_option("gemm_bf16x_variant", variant)
    print(f"gemm_bf16x_variant={variant}", flush=True)
    for N in (320, 960, 2560):
        for tile in (100, 101):
            ts = []
            for K in (64, 128, 320, 640, 1280):
                try:
                    ms = sd.bench_conv(32, K, 64, 64, N, k=1, stride=1, upsample2x=0, tile_cfg=tile, splitk=1, iters=6)
                except Exception as e:
                    ms = float("nan")
                ts.append((K, ms))
            fl = lambda K: 2.0 * 131072 * N * K
            (k0, t0), (k1, t1) = ts[0], ts[-1]
            b = (t1 - t0) / (k1 - k0)
            a = t0 - b * k0
            print(f"N={N:5d} tile {tile}: " + "  ".join(f"K={K}: {ms * 1e3:7.1f} us ({fl(K) / ms / 1e9:5.0f} TF)" for K, ms in ts) +
                  f"   | fit: {a * 1e3:6.1f} us fixed + {b * 64e3:6.2f} us per 64 of K (k loop alone: {2.0 * 131072 * N * 64 / (b * 64) / 1e9:5.0f} TF)", flush=True)
    # deep-K 3x3 convolution on the plain tile (a forced tile is not upgraded to the kernel-row form): what persistence does where the k loop dominates
    for (n, cin, hw, cout) in ((32, 320, 64, 320), (32, 640, 32, 640)):
        ms = sd.bench_conv(n, cin, hw, hw, cout, k=3, stride=1, upsample2x=0, tile_cfg=100, splitk=1, iters=6)
        print(f"conv3x3 {n}x{cin}x{hw}x{hw}->{cout} tile 100: {ms * 1e3:7.1f} us ({2.0 * n * hw * hw * cout * cin * 9 / ms / 1e9:5.0f} TF)", flush=True)
sd.close()
