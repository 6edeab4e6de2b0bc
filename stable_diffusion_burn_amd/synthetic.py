"""Synthetic weights and inputs for the SD v1.4 hot path.

No model checkpoint exists in this environment (no network), so every parity
check and every benchmark runs on seeded synthetic weights of the reference's
architecture.  The rules follow BASELINE.md section 3:

* conv / linear weight and bias:  U(-1/sqrt(fan_in), +1/sqrt(fan_in))
* GroupNorm / LayerNorm gamma:    1 + 0.1 * N(0, 1);   beta: 0.1 * N(0, 1)
* initial latent N(0,1) keyed by the *global* image index (so the result of
  image i does not depend on how many GPUs the batch is sharded over)
* text embeddings N(0,1): cond keyed by global image index, uncond shared
* alphas_cumprod = cumprod(1 - linspace(sqrt(0.00085), sqrt(0.012), 1000)^2)
  (the SD "scaled-linear" schedule; real runs read it from the weight file,
  reference src/model/stablediffusion/load.rs:21)

Every tensor is generated independently from (seed, crc32(name)) so a block
can be instantiated without touching the other ~3.6 GB of parameters.
Tensor names are the reference's npy-dump tree paths
(src/model/unet/load.rs:217-305, src/model/autoencoder/load.rs:135-157).
"""
from __future__ import annotations

import zlib

import numpy as np

WEIGHT_SEED = 3
LATENT_SEED = 0
COND_SEED = 1
UNCOND_SEED = 2


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


class SyntheticWeights:
    """name -> float32 ndarray, generated lazily and deterministically.

    ``get(name, shape, kind, fan_in)`` with kind in {"w", "b", "gamma", "beta"}.
    Both the oracle and the HIP engine pull their parameters through this one
    object, asking by the reference's tensor names and the reference's shapes
    (Linear weight is [in, out] as in python/save.py:19; Conv2d weight is
    [Cout, Cin, kh, kw]).
    """

    def __init__(self, seed: int = WEIGHT_SEED, cache: bool = False):
        self.seed = seed
        self._cache = {} if cache else None

    def get(self, name: str, shape, kind: str, fan_in: int = 0) -> np.ndarray:
        key = (name, tuple(shape), kind, fan_in)
        if self._cache is not None and key in self._cache:
            return self._cache[key]
        g = _rng(self.seed, name)
        if kind in ("w", "b"):
            bound = 1.0 / np.sqrt(float(fan_in))
            a = g.random(tuple(shape), dtype=np.float32)
            a = (a * np.float32(2.0) - np.float32(1.0)) * np.float32(bound)
        elif kind == "gamma":
            a = np.float32(1.0) + np.float32(0.1) * g.standard_normal(tuple(shape), dtype=np.float32)
        elif kind == "beta":
            a = np.float32(0.1) * g.standard_normal(tuple(shape), dtype=np.float32)
        else:
            raise ValueError(f"unknown synthetic kind {kind!r}")
        a = np.ascontiguousarray(a, dtype=np.float32)
        if self._cache is not None:
            self._cache[key] = a
        return a


def named_tensor(provider, name: str, shape, shapes) -> np.ndarray:
    """The synthetic tensor of dump name `name` in the dump's own layout: kind and fan-in follow from the
    tensor's rank and its module's weight shape (`shapes`: name -> shape of every tensor of the model)."""
    parent, leaf = name.rsplit("/", 1)
    wshape = shapes.get(parent + "/weight")
    if leaf == "weight":
        if len(shape) == 4:
            return provider.get(name, shape, "w", shape[1] * shape[2] * shape[3])
        if len(shape) == 2:
            return provider.get(name, shape, "w", shape[0])   # Linear [in, out] / embedding table [rows, width]
        return provider.get(name, shape, "gamma")
    if wshape is not None and len(wshape) == 4:
        return provider.get(name, shape, "b", wshape[1] * wshape[2] * wshape[3])
    if wshape is not None and len(wshape) == 2:
        return provider.get(name, shape, "b", wshape[0])
    return provider.get(name, shape, "beta")


def alphas_cumprod(n: int = 1000) -> np.ndarray:
    """SD scaled-linear schedule, float32 like the reference's weight file."""
    betas = np.linspace(np.sqrt(0.00085), np.sqrt(0.012), n, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas).astype(np.float32)


def initial_latent(global_index: int, h: int = 64, w: int = 64) -> np.ndarray:
    """x_T for image ``global_index``: [4, h, w] float32 NCHW."""
    return _rng(LATENT_SEED, f"latent/{global_index}").standard_normal((4, h, w), dtype=np.float32)


def cond_context(global_index: int, t: int = 77, dim: int = 768) -> np.ndarray:
    """"Random text embedding" standing in for CLIP(prompt): [t, dim]."""
    return _rng(COND_SEED, f"context/{global_index}").standard_normal((t, dim), dtype=np.float32)


def uncond_context(t: int = 77, dim: int = 768) -> np.ndarray:
    return _rng(UNCOND_SEED, "unconditional_context").standard_normal((t, dim), dtype=np.float32)
