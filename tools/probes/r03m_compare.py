"""Host side of tools/probes/r03m_dump.py: the fp64 oracle's half of the precision = 1 / 2 model-level comparisons, evaluated off the GPU
box on the outputs the GPU run saved under gpurun_out/r03m/.  Prints the numbers the bars in tests/test_fp8_gpu.py and
tests/test_golden_gpu.py are set from (bar = 1.5 x measured)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mx_oracle as MX                                          # noqa: E402
from oracle import sd_oracle as O                                           # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn                     # noqa: E402

out = ROOT / "gpurun_out" / "r03m"
cache = Path("/tmp/r03m_oracle_unet8.npz")


def rel_rms(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))


if not cache.exists():
    dims = O.Dims(320, 8, 768, 8, 8, 64)
    lat = torch.from_numpy(np.stack([syn.initial_latent(i, 8, 8) for i in range(2)]))
    ctx = torch.from_numpy(np.stack([syn.cond_context(i, 77, 768) for i in range(2)]))
    res = {"exact": O.UNetOracle(syn.SyntheticWeights(), dims, torch.float64).forward(lat, 999, ctx).numpy()}
    for wide in (0, 1):
        with MX.MxResConvs(wide=bool(wide)):
            res[f"q{wide}"] = O.UNetOracle(syn.SyntheticWeights(), dims, torch.float64).forward(lat, 999, ctx).numpy()
    np.savez(cache, **res)
o = np.load(cache)
for wide in (0, 1):
    print(f"unet8 oracle: format alone wide={wide}: {rel_rms(o[f'q{wide}'], o['exact']):.3e}")
if not (out / "unet8_wide0.npy").exists():
    sys.exit(0)
bf = np.load(out / "unet8_bf16.npy")
print(f"unet8 bf16 (fp8_convs=0) vs exact: {rel_rms(bf, o['exact']):.3e}")
for wide in (0, 1):
    got = np.load(out / f"unet8_wide{wide}.npy")
    print(f"unet8 wide={wide}: vs exact {rel_rms(got, o['exact']):.3e}; vs same quantisation {rel_rms(got, o[f'q{wide}']):.3e}; "
          f"format alone {rel_rms(o[f'q{wide}'], o['exact']):.3e}")

g = np.load(ROOT / "tests" / "golden" / "sd14_synth_cfg5.npz")
for wide in (1, 0):
    got, rgb = np.load(out / f"cfg5_latent_wide{wide}.npy"), np.load(out / f"cfg5_rgb_s4_wide{wide}.npy")
    for i in range(2):
        print(f"cfg5 wide={wide} sample {i}: latent vs exact {rel_rms(got[i], g['latent64'][i]):.3e}, vs fp64 with the ResBlock-conv quantisation "
              f"{rel_rms(got[i], g['latent64_mx'][i]):.3e} (format alone {rel_rms(g['latent64_mx'][i], g['latent64'][i]):.3e}); "
              f"decode of the exact latent, RGB {rel_rms(rgb[i], g['rgb64_s4'][i]):.3e}")
g3 = np.load(ROOT / "tests" / "golden" / "sd14_synth_cfg3.npz")
got, rgb = np.load(out / "cfg3_latent_bf16.npy"), np.load(out / "cfg3_rgb_s4_bf16.npy")
for i in range(2):
    print(f"cfg3 bf16 B=16 S=50 sample {i}: latent vs fp64 {rel_rms(got[i], g3['latent64'][i]):.3e}; decode of the fp64 latent, RGB "
          f"{rel_rms(rgb[i], g3['rgb64_s4'][i]):.3e}")
