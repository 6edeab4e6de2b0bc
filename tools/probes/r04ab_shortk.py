"""Short-K Linear layers in bf16 (M = 131 072 = the 64x64 level at CFG batch 32): time against K at fixed M, N and tile -> fixed cost per output tile (prologue, epilogue, store)
and the k loop's rate; 1x1 convolutions through bench_conv."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
for N in (320, 960, 2560):
    for tile in (100, 101, 103):
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
sd.close()
