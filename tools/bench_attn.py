"""A/B the attention kernels on the UNet's self/cross-attention shapes (HIP events inside libsdmi)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from stable_diffusion_burn_amd import ModelConfig, StableDiffusion  # noqa: E402

SHAPES = [(2, 4096, 4096, 320, 8), (2, 1024, 1024, 640, 8), (2, 256, 256, 1280, 8), (2, 64, 64, 1280, 8),
          (2, 4096, 77, 320, 8), (2, 1024, 77, 640, 8), (2, 256, 77, 1280, 8)]
BF16 = "--bf16" in sys.argv
if "--b16" in sys.argv:
    SHAPES = [(16,) + s[1:] for s in SHAPES]
sd = StableDiffusion(ModelConfig(64, 1, 64, 8, 8, 64, precision=1 if BF16 else 0))
for s in SHAPES:
    n, nq, nk, c, h = s
    fl = 4.0 * n * h * nq * nk * (c // h)
    row = []
    for variant in (0, 1):
        sd.set_option("attn_bf16" if BF16 else "attn_split", variant)   # fp32: 0 = k_attn.hip (fp32 MFMA), 1 = k_attn_split.hip
        ms = sd.bench_attention(*s, iters=10)
        row.append(f"v{variant}: {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF")
    if BF16:   # --bf16-variants 1,2,..: k_attn_bf16.hip forms (option attn_bf16_variant), after the default (0)
        for arg in sys.argv:
            if arg.startswith("--bf16-variants="):
                for bv in arg.split("=", 1)[1].split(","):
                    sd.set_option("attn_bf16_variant", int(bv))
                    ms = sd.bench_attention(*s, iters=10)
                    row.append(f"variant {bv}: {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF")
                sd.set_option("attn_bf16_variant", 0)
    print(f"{str(s):34s} " + " | ".join(row), flush=True)
