"""Per-shape time of the bf16 GEMMs at batch 16 with the engine's own tile choice (where do configs[2]'s 82 ms/image of GEMM go?)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1))
rows = []
for ln in Path("profiles/r01_gemm_shapes_b1.txt").read_text().splitlines():
    key, cnt = ln.split()
    s = tuple(int(v) for v in key.split(","))
    if s[1] % 64 or s[3] in (1, 154) or (s[2] == 1 and s[3] == 1):
        continue
    unet = s[0] == 2 or (s[2] == 1 and s[3] in (8192, 2048, 512, 128))
    n = s[0] * 16 if s[0] == 2 else (16 if s[2] > 1 else 1)
    h, w = (s[2], s[3]) if s[2] > 1 else (1, s[3] * 16)
    sh = (n, s[1], h, w, s[4], s[5], s[6], s[7])
    try:
        ms = sd.bench_conv(*sh[:5], k=sh[5], stride=sh[6], upsample2x=sh[7], tile_cfg=-1, splitk=0, iters=3)
    except Exception as e:  # noqa: BLE001
        print(sh, e)
        continue
    hin, win = h << s[7], w << s[7]
    pad = 1 if s[5] == 3 else 0
    ho, wo = (hin + 2 * pad - s[5]) // s[6] + 1, (win + 2 * pad - s[5]) // s[6] + 1
    fl = 2.0 * n * ho * wo * s[4] * s[1] * s[5] * s[5]
    per_img = ms * int(cnt) * (20 if unet else 1) / 16.0
    rows.append((per_img, sh, int(cnt), ms * 1e3, fl / ms / 1e9))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
for r in rows[:40]:
    print(f"{r[0]:6.2f} ms/img {100 * r[0] / tot:5.1f} %  {str(r[1]):44s} x{r[2]:<3d} {r[3]:8.1f} us {r[4]:7.1f} TF", flush=True)
print("total", tot)
