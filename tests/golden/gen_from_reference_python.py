"""Pin the oracle against the REFERENCE'S OWN Python model + exporters (run in this container only).

    python tests/golden/gen_from_reference_python.py        # ~3-4 min, ~15 GB RAM, 4 GB of /tmp

The Rust reference cannot be built here, but its Python side can be imported: python/dump.py is
the tinygrad model definition the Rust port mirrors, and python/{save,unet,autoencoder}.py are the
exporters that DEFINE the npy-dump tree the Rust loaders read (src/model/load.rs:17-160).  With
the tinygrad-API shim (tests/golden/tinygrad_shim.py) this script

 1. instantiates dump.UNetModel() and dump.AutoencoderKL() (full SD v1.4 size), every parameter
    initialised to a unique constant;
 2. runs the reference's exporters (unet.save_unet_model, autoencoder.save_autoencoder) -> a dump
    tree; the unique constants map every dump file back to the Python attribute it came from;
 3. overwrites every parameter with the seeded synthetic tensor of ITS DUMP NAME
    (stable_diffusion_burn_amd.synthetic.SyntheticWeights; Linear weights transposed exactly as
    save.py:19 does), so the Python model now holds the very weights the oracle and the HIP engine
    generate from names -- the name mapping is the reference's, not ours;
 4. runs the reference Python forward passes (gelu = exact erf as in the Rust code, and once with
    tinygrad's tanh form to size quirk Q4) and stores the outputs as small fixtures:
      refpy_unet.npz     eps of x_T / cond context at t = 999 (same inputs as sd14_synth_unet.npz)
                         + the commented probe of dump.py:622-634 (zeros latent, context
                         [0.5]*384+[1.3]*384, timestep 1.0)
      refpy_decoder.npz  decoder output (stride-4 grid) of the golden final latent / 0.18215
 5. re-exports a few small modules with the synthetic weights through the reference's exporters into
    tests/golden/refdump/ (byte-exact reference-written .npy files: the fixture of the dump reader).

tests/test_reference_python_cpu.py then checks oracle == fixtures WITHOUT needing /root/reference.
"""
import shutil
import sys
import tempfile
import time
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))
REF_PY = Path("/root/reference/python")

import tinygrad_shim as shim  # noqa: E402

from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402


def collect_params(obj, seen, out):
    """All shim Tensors reachable from a model object (lists / dicts / tuples / attributes)."""
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, shim.Tensor):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            collect_params(o, seen, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            collect_params(o, seen, out)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (type, types.FunctionType, types.MethodType, types.ModuleType)):
        for o in vars(obj).values():
            collect_params(o, seen, out)


def read_dump_tensor(path):
    raw = np.load(path)
    return raw


def synth_for(name, shape, shapes, W):
    parent, leaf = name.rsplit("/", 1)
    wshape = shapes.get(parent + "/weight")
    if leaf == "weight":
        if len(shape) == 4:
            return W.get(name, shape, "w", shape[1] * shape[2] * shape[3])
        if len(shape) == 2:
            return W.get(name, shape, "w", shape[0])
        return W.get(name, shape, "gamma")
    if wshape is not None and len(wshape) == 4:
        return W.get(name, shape, "b", wshape[1] * wshape[2] * wshape[3])
    if wshape is not None and len(wshape) == 2:
        return W.get(name, shape, "b", wshape[0])
    return W.get(name, shape, "beta")


def main():
    t0 = time.time()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    shim.install()
    sys.path.insert(0, str(REF_PY))
    import dump  # the reference's model definition (python/dump.py)
    import autoencoder as ae_save  # the reference's exporters
    import unet as unet_save

    unet = dump.UNetModel()
    ae = dump.AutoencoderKL()
    params = []
    collect_params(unet, set(), params)
    n_unet = len(params)
    collect_params(ae, set(), params)
    by_index = {int(p.t.flatten()[0].item()): p for p in params}
    print(f"python model: {n_unet} UNet tensors, {len(params) - n_unet} autoencoder tensors ({time.time() - t0:.0f} s)", flush=True)

    tmp = Path(tempfile.mkdtemp(prefix="refdump_"))
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            unet_save.save_unet_model(unet, tmp / "unet")
            ae_save.save_autoencoder(ae, tmp / "autoencoder")
        print(f"reference exporters wrote the dump tree ({time.time() - t0:.0f} s)", flush=True)

        # map dump files -> python tensors through the unique constants
        mapping = {}   # dump name -> (python tensor, transposed?, shape in dump)
        shapes = {}
        for f in sorted(tmp.rglob("*.npy")):
            name = str(f.relative_to(tmp))[:-4]
            leaf = name.rsplit("/", 1)[1]
            if leaf not in ("weight", "bias"):
                continue
            raw = np.load(f, mmap_mode="r")
            # first D values are the shape (save.py:10-15); D is unknown here: infer from the tensor
            for d in (1, 2, 4):
                dims = [int(v) for v in raw[:d]]
                if len(raw) == d + int(np.prod(dims)) and all(v > 0 for v in dims):
                    break
            else:
                raise RuntimeError(f"cannot parse {f}")
            idx = int(raw[d])
            p = by_index[idx]
            transposed = len(dims) == 2  # every 2-D tensor on this path is a Linear weight: save.py:19 transposes it
            assert tuple(p.shape) == (tuple(dims[::-1]) if transposed else tuple(dims)), (name, p.shape, dims)
            mapping[name] = (p, transposed, tuple(dims))
            shapes[name] = tuple(dims)
        print(f"mapped {len(mapping)} dump tensors to python attributes ({time.time() - t0:.0f} s)", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    # overwrite every parameter with the synthetic tensor of its dump name
    W = syn.SyntheticWeights()
    for name, (p, transposed, dims) in mapping.items():
        if "/encoder/" in name or "quant_conv" in name and "post_quant" not in name:
            arr = np.zeros(dims, np.float32)  # VAE encoder: off the hot path, never evaluated here
        else:
            arr = synth_for(name, dims, shapes, W)
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(shim.DTYPE)
        p.t = t.t().contiguous() if transposed else t
    print(f"synthetic weights installed by dump name ({time.time() - t0:.0f} s)", flush=True)

    out_unet = {}
    x = shim.Tensor(syn.initial_latent(0)[None])
    ctx = shim.Tensor(syn.cond_context(0)[None])
    for mode in ("erf", "tanh"):
        shim.GELU_MODE = mode
        eps = unet(x, shim.Tensor([999.0]), ctx).numpy()[0]
        out_unet[f"eps_t999_{mode}"] = eps
        print(f"reference python UNet forward ({mode}) done ({time.time() - t0:.0f} s)", flush=True)
    shim.GELU_MODE = "erf"
    probe_ctx = np.expand_dims(np.repeat(np.array([0.5, 1.3], dtype=np.float32), 768 // 2), axis=(0, 1))  # dump.py:626-629
    out_unet["probe_zeros"] = unet(shim.Tensor.zeros([1, 4, 64, 64]), shim.Tensor([1.0]), shim.Tensor(probe_ctx)).numpy()[0]
    gold = np.load(HERE / "sd14_synth_unet.npz")
    d_erf = np.abs(out_unet["eps_t999_erf"] - gold["eps64_t999"]).max()
    d_tanh = np.abs(out_unet["eps_t999_tanh"] - gold["eps64_t999"]).max()
    print(f"reference-python (erf) vs oracle f64: max|d| = {d_erf:.3e}; tinygrad tanh-GELU vs oracle: {d_tanh:.3e}", flush=True)
    np.savez_compressed(HERE / "refpy_unet.npz", **{k: v.astype(np.float64) for k, v in out_unet.items()},
                        vs_oracle_erf=np.array(d_erf), vs_oracle_tanh=np.array(d_tanh))

    # decoder: post_quant_conv + decoder of the golden final latent
    g2 = np.load(HERE / "sd14_synth_cfg2.npz")
    z = shim.Tensor((g2["latent64"][None] * (1.0 / 0.18215)))
    img = ae.decoder(ae.post_quant_conv(z)).numpy()[0]
    d_dec = np.abs(img[:, ::4, ::4] - g2["rgb64_s4"]).max()
    print(f"reference-python decoder vs oracle f64 (stride-4 grid): max|d| = {d_dec:.3e} ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(HERE / "refpy_decoder.npz", rgb_s4=img[:, ::4, ::4].astype(np.float64), vs_oracle=np.array(d_dec))

    # byte-exact reference-written dump files of a few small modules (reader fixture)
    ref = HERE / "refdump"
    shutil.rmtree(ref, ignore_errors=True)
    import save as ref_save
    with contextlib.redirect_stdout(io.StringIO()):
        ref_save.save_conv2d(unet.input_blocks[0][0], ref / "unet/input_blocks/conv")                  # Conv2d 4->320
        ref_save.save_linear(unet.time_embed[0], ref / "unet/lin1_time_embed")                        # Linear 320->1280 (transposed)
        ref_save.save_group_norm(unet.out[0], ref / "unet/norm_out")                                  # GroupNorm 320
        ref_save.save_conv2d(unet.out[2], ref / "unet/conv_out")                                      # Conv2d 320->4
        ref_save.save_conv2d(ae.post_quant_conv, ref / "autoencoder/post_quant_conv")                 # 1x1 4->4
        ref_save.save_layer_norm(unet.input_blocks[1][1].transformer_blocks[0].norm1,
                                 ref / "unet/input_blocks/rt1/transformer/transformer/norm1")         # LayerNorm 320
    # float64 shim tensors -> save.py casts to float32 itself (np.concatenate(...).astype(np.float32))
    n_files = len(list(ref.rglob("*.npy")))
    print(f"wrote {n_files} reference-format files under {ref} ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
