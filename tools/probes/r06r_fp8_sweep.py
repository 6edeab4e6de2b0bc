"""round 6, call r: every MXFP8 convolution shape of the precision-2 batch-16 image (from profiles/r06m_shape_times_fp8_b16.txt) under each tile x split-K, HBM-cold,
against the engine's plan (Engine::launch_fp8): where does the plan lose more than 4 %?"""
import math, re, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=2))
sd.set_option("bench_cold", 1)
shapes = []
for ln in Path("profiles/r06m_shape_times_fp8_b16.txt").read_text().splitlines():
    m = re.search(r"gemm_fp8 (\d+),(\d+),(\d+) k3", ln)
    if not m:
        continue
    M, N, K = (int(v) for v in m.groups())
    cp = K // 9
    cin = {384: 320, 1024: 960}.get(cp, cp)
    n = 32 if M % 32 == 0 and math.isqrt(M // 32) ** 2 == M // 32 and M // 32 <= 4096 and N >= 320 else 1
    hw = math.isqrt(M // n)
    launches = float(ln.split()[3])
    shapes.append(((n, cin, hw, hw, N), launches, (M, N, K)))
tot_auto = tot_best = 0.0
for shape, launches, mnk in shapes:
    auto = sd.bench_conv(*shape, k=3, stride=1, upsample2x=0, tile_cfg=-1, splitk=0, iters=4)
    best = (auto, "auto")
    for tile in (0, 1, 2):
        for sp in (1, 2, 3, 4, 6, 8):
            try:
                ms = sd.bench_conv(*shape, k=3, stride=1, upsample2x=0, tile_cfg=tile, splitk=sp, iters=4)
            except Exception:  # noqa: BLE001
                continue
            if ms < best[0]:
                best = (ms, f"t{tile}x{sp}")
    tot_auto += auto * launches; tot_best += best[0] * launches
    print(f"{str(shape):34s} M,N,K={mnk}  x{launches:5.1f}/img  plan {auto * 1e3:7.1f} us  best {best[0] * 1e3:7.1f} ({best[1]})" + ("   <-- > 4 %" if best[0] < 0.96 * auto else ""), flush=True)
print(f"launch-weighted per image: plan {tot_auto:.3f} ms, best {tot_best:.3f} ms")
