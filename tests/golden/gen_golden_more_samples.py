"""Golden fixture, round 5: MORE SAMPLES of the batched configurations (BASELINE.json configs[2..4]) -- the verdict of round 4 noted that the batch-16 tests compared
samples 0 and 1 only, and that the per-GPU shard of configs[3] (B = 8, 20 steps, bf16) had no golden test of its own.

    python tests/golden/gen_golden_more_samples.py [threads]      # ~50 min on 6 threads (fp64 oracle)

Batch > 1 = independent batch-1 samples: global image indices 7 and 15 (x_T keyed by the index, one prompt embedding for all -- what bench.py and
the GPU tests feed), each run as a batch-1 sample by the fp64 oracle:

  sd14_synth_more.npz
       index           [2]            i64   = [7, 15]
       latent64_s20    [2,4,64,64]    f64   exact network, 20-step schedule t = 999, 949, .., 49  (configs[3] shard: sample 7 is its last image; configs[4] shard)
       rgb64_s20_s4    [2,3,128,128]  f64   decoded float RGB of latent64_s20 on a stride-4 grid
       latent64_mx_s20 [2,4,64,64]    f64   the network with MXFP8 ResBlock / ResnetBlock 3x3 convolutions (oracle/mx_oracle.py MxResConvs), 20 steps
       latent64_s50    [2,4,64,64]    f64   exact network, 50-step schedule t = 999, 979, .., 19  (configs[2])
       rgb64_s50_s4    [2,3,128,128]  f64   decoded float RGB of latent64_s50 on a stride-4 grid

Samples 0 and 1 of the same runs are in sd14_synth_cfg3.npz (50 steps) and sd14_synth_cfg5.npz (20 steps, exact and MX).
PARITY UNPINNED beyond the oracle's own pinning (oracle/sd_oracle.py header).  Nothing here reads /root/reference.
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
INDEX = [7, 15]


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
    d = Dims()
    w = syn.SyntheticWeights(cache=True)
    a = syn.alphas_cumprod()
    ctx = torch.from_numpy(syn.cond_context(0))[None]
    unc = torch.from_numpy(syn.uncond_context())
    t0 = time.time()
    out = {k: [] for k in ("latent64_s20", "rgb64_s20_s4", "latent64_mx_s20", "latent64_s50", "rgb64_s50_s4")}
    part = OUT / "sd14_synth_more.partial.npz"
    for i in INDEX:
        x = torch.from_numpy(syn.initial_latent(i))[None]
        o64 = StableDiffusionOracle(w, a, d, torch.float64)
        l20 = o64.sample_latent(ctx, unc, 7.5, 20, x)
        out["latent64_s20"].append(l20.numpy()[0])
        out["rgb64_s20_s4"].append(o64.decode_float(l20)[0].numpy()[:, ::4, ::4].copy())
        print(f"sample {i} exact 20 steps: latent absmax {np.abs(out['latent64_s20'][-1]).max():.2f} ({time.time() - t0:.0f} s)", flush=True)
        with MX.MxResConvs():
            oq = StableDiffusionOracle(w, a, d, torch.float64)
            out["latent64_mx_s20"].append(oq.sample_latent(ctx, unc, 7.5, 20, x).numpy()[0])
        r = float(np.sqrt(np.mean((out["latent64_mx_s20"][-1] - out["latent64_s20"][-1]) ** 2) / np.mean(out["latent64_s20"][-1] ** 2)))
        print(f"sample {i} MXFP8 ResBlock convs, 20 steps: rel-RMS vs exact {r:.3e} ({time.time() - t0:.0f} s)", flush=True)
        l50 = o64.sample_latent(ctx, unc, 7.5, 50, x)
        out["latent64_s50"].append(l50.numpy()[0])
        out["rgb64_s50_s4"].append(o64.decode_float(l50)[0].numpy()[:, ::4, ::4].copy())
        print(f"sample {i} exact 50 steps: latent absmax {np.abs(out['latent64_s50'][-1]).max():.2f} ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(part, index=np.array(INDEX[:len(out["latent64_s50"])], dtype=np.int64), **{k: np.stack(v) for k, v in out.items()})
    np.savez_compressed(OUT / "sd14_synth_more.npz", index=np.array(INDEX, dtype=np.int64), **{k: np.stack(v) for k, v in out.items()})
    part.unlink(missing_ok=True)
    print(f"done ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
