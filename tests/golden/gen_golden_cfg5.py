"""Golden fixture for BASELINE.json configs[4] (B = 128 over 8 GPUs = 16 per GPU, 20 DDIM steps, CFG 7.5, "fp8 conv+attn").

    python tests/golden/gen_golden_cfg5.py        # ~35 min on 6 threads (fp64 oracle: 2 x 20 CFG steps, twice)

Batch > 1 = independent batch-1 samples, so the fixture holds the fp64 oracle's result for the first TWO samples of a GPU's
16 (global image indices 0 and 1; x_T keyed by the index, one prompt embedding for all), each run as a batch-1 sample, in two forms:

  sd14_synth_cfg5.npz
       latent64     [2,4,64,64]  f64   the exact network (no quantisation): what every precision is measured against
       latent64_mx  [2,4,64,64]  f64   the network with its ResBlock / ResnetBlock 3x3 convolutions taking MXFP8 inputs and weights
                                       (oracle/mx_oracle.py MxResConvs: the quantisation precision = 2 applies), fp64 everywhere else
       rgb64_s4     [2,3,128,128] f64  decoded float RGB of `latent64` on a stride-4 grid (exact decoder)
       rgb64_mx_s4  [2,3,128,128] f64  decoded float RGB of `latent64_mx` through the decoder WITH the MXFP8 ResnetBlock convolutions
       timesteps    [20]          i64

The reference has no reduced-precision arithmetic (src/bin/sample/main.rs:59-64); the MX rules are the OCP specification's, pinned to
the hardware's conversions in tests/test_mx_oracle_cpu.py.  Nothing here reads /root/reference.
"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import mx_oracle as MX  # noqa: E402
from oracle.sd_oracle import Dims, StableDiffusionOracle  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
    d = Dims()
    w = syn.SyntheticWeights(cache=True)
    a = syn.alphas_cumprod()
    ctx = torch.from_numpy(syn.cond_context(0))[None]
    unc = torch.from_numpy(syn.uncond_context())
    t0 = time.time()
    out = {"latent64": [], "latent64_mx": [], "rgb64_s4": [], "rgb64_mx_s4": []}
    for i in range(2):
        x = torch.from_numpy(syn.initial_latent(i))[None]
        o64 = StableDiffusionOracle(w, a, d, torch.float64)
        l64 = o64.sample_latent(ctx, unc, 7.5, 20, x)
        out["latent64"].append(l64.numpy()[0])
        out["rgb64_s4"].append(o64.decode_float(l64)[0].numpy()[:, ::4, ::4].copy())
        print(f"sample {i} exact: latent absmax {np.abs(out['latent64'][-1]).max():.2f} ({time.time() - t0:.0f} s)", flush=True)
        with MX.MxResConvs():
            oq = StableDiffusionOracle(w, a, d, torch.float64)
            lq = oq.sample_latent(ctx, unc, 7.5, 20, x)
            out["latent64_mx"].append(lq.numpy()[0])
            out["rgb64_mx_s4"].append(oq.decode_float(lq)[0].numpy()[:, ::4, ::4].copy())
        r = float(np.sqrt(np.mean((out["latent64_mx"][-1] - out["latent64"][-1]) ** 2) / np.mean(out["latent64"][-1] ** 2)))
        print(f"sample {i} MXFP8 ResBlock convs: rel-RMS of the 20-step latent vs exact {r:.3e} ({time.time() - t0:.0f} s)", flush=True)
    ts = np.arange(999, -1, -50, dtype=np.int64)
    np.savez_compressed(OUT / "sd14_synth_cfg5.npz", timesteps=ts, **{k: np.stack(v) for k, v in out.items()})
    print(f"done ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
