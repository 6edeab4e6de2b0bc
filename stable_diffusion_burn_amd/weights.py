"""The reference's npy-dump weight format, reader and writer.

Format (defined by the reference's exporters python/save.py:6-68 and read by
src/model/load.rs:17-160): every tensor is a 1-D float32 `.npy` whose first D values are the
shape and whose remaining values are the row-major data; a scalar s is stored as `[1.0, s]`;
Linear `weight` is stored TRANSPOSED to [in, out] (save.py:19); a Conv2d directory holds
`weight [Cout,Cin,kh,kw]`, `bias`, and the 2-vectors `stride`, `padding`, `dilation`,
`kernel_size` plus scalars `n_group`, `n_channels_in`, `n_channels_out`; a GroupNorm directory
holds `weight`, `bias`, `eps`, `n_group`, `n_channel`; a LayerNorm directory `weight`, `bias`, `eps`.
Directory names are the Rust struct field names (src/model/unet/load.rs, autoencoder/load.rs).

The engine's C++ reader (`sdmi_load_weights_dir`, csrc/engine.cpp) reads the `weight` / `bias`
files of the hot-path subset; `write_dump_tree` here writes the COMPLETE per-module file set the
Rust loaders expect, so a tree written from any provider (e.g. synthetic weights) is loadable by
both.  tests/test_reference_python_cpu.py checks the writer byte-for-byte against files produced
by the reference's own exporters (tests/golden/refdump/).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


def encode_tensor(a) -> np.ndarray:
    """save_tensor (python/save.py:10-15): [dims..., values...] as float32."""
    a = np.asarray(a, dtype=np.float32)
    return np.concatenate((np.array(a.shape, dtype=np.float64), a.reshape(-1).astype(np.float64))).astype(np.float32)


def encode_scalar(s) -> np.ndarray:
    """save_scalar (python/save.py:6-8)."""
    return np.array([1.0, float(s)]).astype(np.float32)


def read_tensor(path, ndim: int) -> np.ndarray:
    """numpy_to_tensor (src/model/load.rs:17-28) for a tensor of known rank."""
    raw = np.load(path)
    if raw.ndim != 1 or raw.dtype != np.float32:
        raise ValueError(f"{path}: expected a 1-D float32 array")
    dims = tuple(int(v) for v in raw[:ndim])
    if raw.size != ndim + int(np.prod(dims)):
        raise ValueError(f"{path}: shape prefix {dims} does not match {raw.size - ndim} values")
    return raw[ndim:].reshape(dims)


def _save(path: Path, arr: np.ndarray) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    np.save(path, arr)


def write_conv2d(dirpath, weight, bias, stride=1, padding=0, dilation=1) -> None:
    """save_conv2d (python/save.py:52-68)."""
    d = Path(dirpath)
    weight = np.asarray(weight, np.float32)
    _save(d / "weight.npy", encode_tensor(weight))
    if bias is not None:
        _save(d / "bias.npy", encode_tensor(bias))
    for name, v in (("stride", stride), ("padding", padding), ("dilation", dilation), ("kernel_size", weight.shape[2:])):
        pair = (v, v) if np.isscalar(v) else tuple(v)
        _save(d / f"{name}.npy", encode_tensor(np.array(pair, np.float32)))
    _save(d / "n_group.npy", encode_scalar(1))
    _save(d / "n_channels_in.npy", encode_scalar(weight.shape[1]))
    _save(d / "n_channels_out.npy", encode_scalar(weight.shape[0]))


def write_linear(dirpath, weight_in_out, bias) -> None:
    """save_linear (python/save.py:17-21); `weight_in_out` is already [in, out] (what the file holds)."""
    d = Path(dirpath)
    _save(d / "weight.npy", encode_tensor(weight_in_out))
    if bias is not None:
        _save(d / "bias.npy", encode_tensor(bias))


def write_group_norm(dirpath, gamma, beta, eps=1e-5, n_group=32) -> None:
    """save_group_norm (python/save.py:29-37)."""
    d = Path(dirpath)
    _save(d / "weight.npy", encode_tensor(gamma))
    _save(d / "bias.npy", encode_tensor(beta))
    _save(d / "eps.npy", encode_scalar(eps))
    _save(d / "n_group.npy", encode_scalar(n_group))
    _save(d / "n_channel.npy", encode_scalar(len(gamma)))


def write_layer_norm(dirpath, gamma, beta, eps=1e-5) -> None:
    """save_layer_norm (python/save.py:23-27)."""
    d = Path(dirpath)
    _save(d / "weight.npy", encode_tensor(gamma))
    _save(d / "bias.npy", encode_tensor(beta))
    _save(d / "eps.npy", encode_scalar(eps))


def write_embedding(dirpath, weight) -> None:
    """save_embedding (python/save.py:97-99): the table as it is, [rows, width] (not transposed)."""
    _save(Path(dirpath) / "weight.npy", encode_tensor(weight))


def write_dump_tree(dump_dir, specs, get_tensor, alphas_cumprod, n_head: int = 8, clip_heads: int = 12) -> None:
    """Write the hot-path subset of the dump tree (+ the clip/ subtree when `specs` lists it).

    specs: [(name, shape)] from StableDiffusion.weight_specs(); get_tensor(name, shape) -> ndarray in
    the dump's own layout (Linear [in,out], Conv [Cout,Cin,kh,kw]).
    """
    root = Path(dump_dir)
    shapes = dict(specs)
    modules = {}
    for name, shape in specs:
        if name == "alphas_cumprod":
            continue
        parent, leaf = name.rsplit("/", 1)
        modules.setdefault(parent, {})[leaf] = get_tensor(name, shape)
    for parent, t in modules.items():
        w = t["weight"]
        b = t.get("bias")
        leaf_dir = parent.rsplit("/", 1)[1]
        if parent.startswith("clip/") and leaf_dir.endswith("_embedding"):
            write_embedding(root / parent, w)                                   # python/clip.py:32-35
        elif parent.startswith("clip/") and w.ndim == 1:
            write_layer_norm(root / parent, w, b)                               # attn_ln / mlp_ln / layer_norm
        elif w.ndim == 4 and parent.endswith("/downsampler/conv"):
            write_conv2d(root / parent, w, b, stride=2, padding=0)              # save_padded_conv2d (python/save.py:70-77)
        elif w.ndim == 4:
            k = w.shape[2]
            stride = 2 if parent.rsplit("/", 1)[1] in ("d1", "d2", "d3") else 1   # Downsample (unet/mod.rs:408-427)
            write_conv2d(root / parent, w, b, stride=stride, padding=1 if k == 3 else 0)
        elif w.ndim == 2:
            write_linear(root / parent, w, b)
        elif parent.rsplit("/", 1)[1] in ("norm1", "norm2", "norm3") and "/transformer/transformer/" in parent + "/":
            write_layer_norm(root / parent, w, b)
        else:
            write_group_norm(root / parent, w, b)
        if parent.rsplit("/", 1)[1] in ("attn1", "attn2"):
            pass
    for parent in {p.rsplit("/", 1)[0] for p in modules if p.rsplit("/", 1)[1] in ("query",)}:
        heads = clip_heads if parent.startswith("clip/") else n_head               # unet/load.rs:46, clip/load.rs:33
        _save(root / parent / "n_head.npy", encode_scalar(heads))
    clip_blocks = {p.split("/")[2] for p in modules if p.startswith("clip/blocks/")}
    if clip_blocks:
        _save(root / "clip" / "n_layer.npy", encode_scalar(len(clip_blocks)))      # python/clip.py:29, clip/load.rs:74
    a = np.asarray(alphas_cumprod, np.float32)
    _save(root / "n_steps.npy", encode_scalar(len(a)))                                # stablediffusion/load.rs:20
    _save(root / "alphas_cumprod.npy", encode_tensor(a))
    if any(n.startswith("autoencoder/decoder/blocks/") for n in shapes):
        _save(root / "autoencoder/decoder/n_block.npy", encode_scalar(4))           # autoencoder/load.rs:139
    if any(n.startswith("autoencoder/encoder/blocks/") for n in shapes):
        _save(root / "autoencoder/encoder/n_block.npy", encode_scalar(4))           # autoencoder/load.rs:163
