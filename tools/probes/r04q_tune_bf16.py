"""Sweep the bf16 tiles / split-K per GEMM shape of the batch-8 and batch-16 models (UNet at n = 2 B per forward, VAE at n = 1) against the engine's own choice
(cost model + tuning/gfx950_bf16.txt), operands hot and HBM-cold; prints "M,N,K=cfg,splits" lines for the shapes where the best beats the choice by more than 4 % in BOTH."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
TILES = [0, 3, 7, 9, 100, 101, 102, 103]
seen = {}
for B in (16, 8):
    for ln in Path("profiles/r01_gemm_shapes_b1.txt").read_text().splitlines():
        key, cnt = ln.split()
        s = tuple(int(v) for v in key.split(","))
        if s[1] % 64 or s[3] in (1, 154) or (s[2] == 1 and s[3] == 1):
            continue
        unet = s[0] == 2 or (s[2] == 1 and s[3] in (8192, 2048, 512, 128))
        if not unet:
            if B == 8:
                continue          # the decoder runs one image at a time: the same shapes at every batch
            sh = s
        else:
            n = s[0] * B if s[0] == 2 else 1
            h, w = (s[2], s[3]) if s[2] > 1 else (1, s[3] * B)
            sh = (n, s[1], h, w, s[4], s[5], s[6], s[7])
        hin, win = sh[2] << sh[7], sh[3] << sh[7]
        pad = 1 if sh[5] == 3 else 0
        ho, wo = (hin + 2 * pad - sh[5]) // sh[6] + 1, (win + 2 * pad - sh[5]) // sh[6] + 1
        M, N, K = sh[0] * ho * wo, sh[4], sh[1] * sh[5] * sh[5]
        mk = f"{M},{N},{K}"
        if mk in seen:
            continue
        seen[mk] = 1
        fl = 2.0 * M * N * K
        kt = K // 64
        res = {}
        for cold in (0, 1):
            sd.set_option("bench_cold", cold)
            it = 4 if fl < 5e11 else 2
            auto = sd.bench_conv(*sh[:5], k=sh[5], stride=sh[6], upsample2x=sh[7], tile_cfg=-1, splitk=0, iters=it)
            best = (auto, -1, 0)
            for t in TILES:
                bm, bn = {0: (128, 128), 3: (256, 128), 7: (128, 160), 9: (64, 160), 100: (256, 320), 101: (256, 256), 102: (256, 128), 103: (128, 320)}[t]
                tiles = -(-M // bm) * -(-N // bn)
                for sp in (1, 2, 3, 4, 6, 8):
                    if sp > 1 and (tiles * sp > 1024 or kt // sp < 4 or tiles >= 400):
                        continue
                    try:
                        ms = sd.bench_conv(*sh[:5], k=sh[5], stride=sh[6], upsample2x=sh[7], tile_cfg=t, splitk=sp, iters=it)
                    except Exception:
                        continue
                    if ms < best[0]:
                        best = (ms, t, sp)
            res[cold] = (auto, best)
        (a0, b0), (a1, b1) = res[0], res[1]
        flag = ""
        if b0[1] >= 0 and b0[1] == b1[1] and b0[2] == b1[2] and b0[0] < 0.96 * a0 and b1[0] < 0.96 * a1:
            flag = f"  TUNE {mk}={b0[1]},{b0[2]}"
        print(f"B={B:2d} {str(sh):44s} x{cnt:>3s} {mk:22s} hot: auto {a0 * 1e3:8.1f} us, best {b0[0] * 1e3:8.1f} (tile {b0[1]}, x{b0[2]}) | cold: auto {a1 * 1e3:8.1f}, best {b1[0] * 1e3:8.1f} (tile {b1[1]}, x{b1[2]}){flag}", flush=True)
sd.close()
