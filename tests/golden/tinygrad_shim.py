"""Minimal tinygrad-API shim over torch-CPU -- TEST INFRASTRUCTURE ONLY.

The reference ships a Python definition of the same network (python/dump.py, "the
architecture the Rust port mirrors 1:1") written against tinygrad,
which is not installed here.  This module provides just enough of `tinygrad.tensor.Tensor`,
`tinygrad.nn.{Conv2d,Linear,GroupNorm,LayerNorm,Embedding}` and friends for
`/root/reference/python/dump.py` and the reference's exporters (python/save.py, unet.py,
autoencoder.py) to be IMPORTED AND RUN unmodified by tests/golden/gen_from_reference_python.py.
Op semantics follow tinygrad's documented behaviour (cross-correlation conv, Linear weight
[out,in], biased-variance layernorm with eps inside the sqrt, softmax over the last axis).

`GELU_MODE`: tinygrad's Tensor.gelu() is the tanh approximation, while the Rust reference uses
Burn's exact-erf Gelu.  "erf" (default here) makes the Python model compute
what the Rust model computes; "tanh" reproduces tinygrad literally (used to report the size of
that documented deviation).
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

DTYPE = torch.float64
GELU_MODE = "erf"
_param_counter = [0]


def _unwrap(x):
    return x.t if isinstance(x, Tensor) else x


class Tensor:
    no_grad = True

    def __init__(self, data, dtype=None):
        if isinstance(data, Tensor):
            data = data.t
        if isinstance(data, torch.Tensor):
            self.t = data
        else:
            self.t = torch.as_tensor(np.asarray(data, dtype=np.float64), dtype=DTYPE)

    # ---- construction ----------------------------------------------------------
    @staticmethod
    def zeros(*shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return Tensor(torch.zeros(*shape, dtype=DTYPE))

    @staticmethod
    def empty(*shape):
        return Tensor.zeros(*shape)

    @staticmethod
    def full(shape, value):
        return Tensor(torch.full(tuple(shape), value, dtype=DTYPE))

    @staticmethod
    def arange(n):
        return Tensor(torch.arange(n, dtype=DTYPE))

    # ---- shape ops ---------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self.t.shape)

    def reshape(self, *shape, **kw):
        if "shape" in kw:
            shape = kw["shape"]
        elif len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = shape[0]
        return Tensor(self.t.reshape(*shape))

    def permute(self, *order):
        return Tensor(self.t.permute(*order))

    def expand(self, *shape):
        return Tensor(self.t.expand(*shape))

    def transpose(self, a=1, b=0):
        return Tensor(self.t.transpose(a, b))

    def unsqueeze(self, dim):
        return Tensor(self.t.unsqueeze(dim))

    def chunk(self, n, dim=0):
        return tuple(Tensor(c) for c in self.t.chunk(n, dim=dim))

    def cat(self, *others, dim=0):
        return Tensor(torch.cat([self.t] + [_unwrap(o) for o in others], dim=dim))

    def triu(self, k=0):
        return Tensor(self.t.triu(k))

    def __getitem__(self, idx):
        return Tensor(self.t[idx])

    # ---- math ------------------------------------------------------------------------
    def _bin(self, other, fn):
        return Tensor(fn(self.t, _unwrap(other) if isinstance(other, Tensor) else other))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return Tensor(_unwrap(o) - self.t)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul)
    def __truediv__(self, o): return self._bin(o, torch.div)
    def __neg__(self): return Tensor(-self.t)
    def __matmul__(self, o): return Tensor(self.t @ _unwrap(o))

    def dot(self, o): return Tensor(self.t @ _unwrap(o))
    def exp(self): return Tensor(self.t.exp())
    def cos(self): return Tensor(self.t.cos())
    def sin(self): return Tensor(self.t.sin())
    def sigmoid(self): return Tensor(torch.sigmoid(self.t))
    def softmax(self, axis=-1): return Tensor(torch.softmax(self.t, dim=axis))
    def swish(self): return Tensor(self.t * torch.sigmoid(self.t))
    def silu(self): return self.swish()
    def quick_gelu(self): return Tensor(self.t * torch.sigmoid(1.702 * self.t))

    def gelu(self):
        x = self.t
        if GELU_MODE == "tanh":  # tinygrad: 0.5 * x * (1 + tanh(x * 0.7978845608 * (1 + 0.044715 * x * x)))
            return Tensor(0.5 * x * (1 + torch.tanh(x * 0.7978845608 * (1 + 0.044715 * x * x))))
        return Tensor(0.5 * x * (1 + torch.erf(x / math.sqrt(2.0))))

    def sequential(self, fns):
        x = self
        for f in fns:
            x = f(x)
        return x

    def realize(self): return self
    def numpy(self): return self.t.detach().cpu().numpy()


# ---- nn ---------------------------------------------------------------------------------
def _new_param(*shape):
    """Parameters start as a UNIQUE CONSTANT (their creation index) so that the dump files the
    reference's exporters write can be traced back to the Python attribute they came from."""
    _param_counter[0] += 1
    return Tensor(torch.full(tuple(shape), float(_param_counter[0]), dtype=DTYPE))


class Conv2d:
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        self.weight = _new_param(out_channels, in_channels // groups, *self.kernel_size)
        self.bias = _new_param(out_channels) if bias else None

    def __call__(self, x):
        t = x.t
        pad = self.padding
        if isinstance(pad, (tuple, list)) and len(pad) == 4:  # tinygrad (left, right, top, bottom)
            t = F.pad(t, tuple(pad))
            pad = 0
        return Tensor(F.conv2d(t, self.weight.t, None if self.bias is None else self.bias.t, stride=self.stride,
                               padding=pad, dilation=self.dilation, groups=self.groups))


class Linear:
    def __init__(self, in_features, out_features, bias=True):
        self.weight = _new_param(out_features, in_features)
        self.bias = _new_param(out_features) if bias else None

    def __call__(self, x):
        return Tensor(F.linear(x.t, self.weight.t, None if self.bias is None else self.bias.t))


class GroupNorm:
    def __init__(self, num_groups, num_channels, eps=1e-5, affine=True):
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = _new_param(num_channels) if affine else None
        self.bias = _new_param(num_channels) if affine else None

    def __call__(self, x):
        return Tensor(F.group_norm(x.t, self.num_groups, self.weight.t, self.bias.t, self.eps))


class LayerNorm:
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        self.normalized_shape = (normalized_shape,) if isinstance(normalized_shape, int) else tuple(normalized_shape)
        self.eps = eps
        self.weight = _new_param(*self.normalized_shape)
        self.bias = _new_param(*self.normalized_shape)

    def __call__(self, x):
        return Tensor(F.layer_norm(x.t, self.normalized_shape, self.weight.t, self.bias.t, self.eps))


class Embedding:
    def __init__(self, vocab_size, embed_size):
        self.weight = _new_param(vocab_size, embed_size)

    def __call__(self, idx):
        return Tensor(self.weight.t[idx.t.long()])


def install():
    """Register the shim as `tinygrad` (+ submodules) in sys.modules."""
    root = types.ModuleType("tinygrad")
    tensor_m = types.ModuleType("tinygrad.tensor")
    helpers_m = types.ModuleType("tinygrad.helpers")
    nn_m = types.ModuleType("tinygrad.nn")
    state_m = types.ModuleType("tinygrad.nn.state")
    tensor_m.Tensor = Tensor
    root.Tensor = Tensor
    root.dtypes = types.SimpleNamespace(float32="float32", float16="float16")
    helpers_m.GlobalCounters = types.SimpleNamespace(reset=lambda: None)
    nn_m.Conv2d, nn_m.Linear, nn_m.GroupNorm, nn_m.LayerNorm, nn_m.Embedding = Conv2d, Linear, GroupNorm, LayerNorm, Embedding
    state_m.torch_load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no checkpoint in this environment"))
    state_m.load_state_dict = lambda *a, **k: None
    nn_m.state = state_m
    root.tensor, root.helpers, root.nn = tensor_m, helpers_m, nn_m
    for name, mod in (("tinygrad", root), ("tinygrad.tensor", tensor_m), ("tinygrad.helpers", helpers_m),
                      ("tinygrad.nn", nn_m), ("tinygrad.nn.state", state_m)):
        sys.modules[name] = mod
