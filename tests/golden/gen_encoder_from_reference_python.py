"""Pin the VAE-encoder oracle against the REFERENCE'S OWN Python model (run in this container only).

    python tests/golden/gen_encoder_from_reference_python.py        # ~20 s

Same method as gen_from_reference_python.py, for the encoder half: python/dump.py's
AutoencoderKL through the tinygrad-API shim, dump names from the reference's exporter
(python/autoencoder.py: save_autoencoder), seeded synthetic weights installed by dump name, then
    latent = quant_conv(encoder(x))[:, 0:4]                 (dump.py:144-147 == autoencoder/mod.rs:60-66)
on a seeded 3 x 64 x 64 image (full channel widths, 8 x 8 latent).  Writes tests/golden/refpy_encoder.npz
(input image, latent, and the list of encoder dump names).
"""
import contextlib
import io
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
REF_PY = Path("/root/reference/python")

import tinygrad_shim as shim  # noqa: E402

from gen_from_reference_python import collect_params, synth_for  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402


def main():
    torch.set_num_threads(min(16, torch.get_num_threads()))
    shim.install()
    sys.path.insert(0, str(REF_PY))
    import dump
    import autoencoder as ae_save

    ae = dump.AutoencoderKL()
    params = []
    collect_params(ae, set(), params)
    by_index = {int(p.t.flatten()[0].item()): p for p in params}
    tmp = Path(tempfile.mkdtemp(prefix="refenc_"))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ae_save.save_autoencoder(ae, tmp / "autoencoder")
        mapping, shapes = {}, {}
        for f in sorted(tmp.rglob("*.npy")):
            name = str(f.relative_to(tmp))[:-4]
            if name.rsplit("/", 1)[1] not in ("weight", "bias"):
                continue
            raw = np.load(f, mmap_mode="r")
            for d in (1, 4):
                dims = [int(v) for v in raw[:d]]
                if len(raw) == d + int(np.prod(dims)) and all(v > 0 for v in dims):
                    break
            else:
                raise RuntimeError(f"cannot parse {f}")
            p = by_index[int(raw[d])]
            assert tuple(p.shape) == tuple(dims), (name, p.shape, dims)
            mapping[name] = (p, tuple(dims))
            shapes[name] = tuple(dims)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    enc_names = sorted(n for n in mapping if n.startswith("autoencoder/encoder/") or n.startswith("autoencoder/quant_conv/"))
    print(f"mapped {len(mapping)} autoencoder tensors, {len(enc_names)} on the encoder side")

    W = syn.SyntheticWeights()
    for name in enc_names:
        p, dims = mapping[name]
        p.t = torch.from_numpy(np.ascontiguousarray(synth_for(name, dims, shapes, W))).to(shim.DTYPE)

    img = np.random.default_rng(11).uniform(-1.0, 1.0, (1, 3, 64, 64)).astype(np.float32)
    lat = ae.quant_conv(ae.encoder(shim.Tensor(img)))[:, 0:4].numpy()
    print(f"reference-python encoder: latent {lat.shape}, absmax {np.abs(lat).max():.3f}")
    # the commented probe of dump.py:613-619: autoencoder(Tensor.zeros((1, 3, 10, 10))) -- encode (10 -> 5 -> 3 -> 2), first four
    # moment channels, post_quant_conv, decode (2 -> 16); needs the decoder weights too
    for name in mapping:
        if name not in enc_names:
            p, dims = mapping[name]
            p.t = torch.from_numpy(np.ascontiguousarray(synth_for(name, dims, shapes, W))).to(shim.DTYPE)
    probe = ae(shim.Tensor(np.zeros((1, 3, 10, 10), np.float32))).numpy()
    print(f"reference-python autoencoder probe: output {probe.shape}, absmax {np.abs(probe).max():.3f}")
    np.savez_compressed(HERE / "refpy_encoder.npz", image=img, latent=lat.astype(np.float64), dump_names=np.array(enc_names),
                        probe_zeros_10x10=probe.astype(np.float64))
    print("wrote refpy_encoder.npz")


if __name__ == "__main__":
    main()
