"""CPU ORACLE for the CLIP text encoder -- TEST INFRASTRUCTURE ONLY.

Restatement of src/model/clip/mod.rs (CLIP::forward :56-75, ResidualDecoderAttentionBlock :110-114,
MultiHeadSelfAttention :158-180, MLP/QuickGELU :207-226) and attn_decoder_mask (src/backend.rs:130-139),
on torch-CPU in fp32 / fp64, with weights addressed by the reference's dump names
(src/model/clip/load.rs:14-91; written by python/clip.py).

PARITY: pinned to the reference's Python model -- tests/golden/gen_clip_from_reference_python.py runs
python/dump.py's CLIPTextTransformer (through the tinygrad-API shim) on the same seeded weights,
installed by the dump names the reference's own exporter (python/clip.py) assigns; the outputs are
committed as tests/golden/refpy_clip.npz and re-checked by tests/test_clip_cpu.py.  Burn's kernels
themselves are not run (no Rust toolchain), as for the rest of the oracle (see sd_oracle.py).

Only tests/ may import this module; the product never does.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import sd_oracle as O


@dataclass(frozen=True)
class ClipDims:
    """CLIPConfig::new(49408, 768, 12, 77, 12) -- stablediffusion/mod.rs:29."""
    n_vocab: int = 49408
    n_state: int = 768
    n_head: int = 12
    n_ctx: int = 77
    n_layer: int = 12


def attn_decoder_mask(seq_len: int, dtype=torch.float32) -> torch.Tensor:
    """src/backend.rs:130-139: zeros with -inf strictly above the diagonal."""
    mask = torch.zeros((seq_len, seq_len), dtype=dtype)
    for i in range(seq_len - 1):
        mask[i, i + 1:] = float("-inf")
    return mask


def quick_gelu(x):
    """clip/mod.rs:223-225: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(x * 1.702)


class CLIPOracle:
    def __init__(self, provider, dims: ClipDims = ClipDims(), dtype=torch.float32, root="clip"):
        self.P = O.Params(provider, dtype)
        self.d = dims
        self.dtype = dtype
        self.root = root

    def _table(self, name, rows):
        return self.P._t(self.P.p.get(f"{self.root}/{name}/weight", (rows, self.d.n_state), "w", rows))

    def block(self, path, x, mask):
        """ResidualDecoderAttentionBlock::forward, clip/mod.rs:110-114."""
        c = self.d.n_state
        h = O.layer_norm(x, *self.P.norm(f"{path}/attn_ln", c))
        q = O.linear(h, *self.P.linear(f"{path}/attn/query", c, c))       # clip/mod.rs:159-161 (all with bias)
        k = O.linear(h, *self.P.linear(f"{path}/attn/key", c, c))
        v = O.linear(h, *self.P.linear(f"{path}/attn/value", c, c))
        wv = O.qkv_attention(q, k, v, mask, self.d.n_head)                 # :171-177
        x = x + O.linear(wv, *self.P.linear(f"{path}/attn/out", c, c))
        h = O.layer_norm(x, *self.P.norm(f"{path}/mlp_ln", c))
        h = O.linear(h, *self.P.linear(f"{path}/mlp/fc1", c, 4 * c))       # MLP::forward :207-213
        h = quick_gelu(h)
        return x + O.linear(h, *self.P.linear(f"{path}/mlp/fc2", 4 * c, c))

    def forward(self, tokens) -> torch.Tensor:
        """CLIP::forward, clip/mod.rs:56-75: tokens int [n, T] -> [n, T, n_state]."""
        tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
        n, T = tokens.shape
        if T > self.d.n_ctx:
            raise ValueError("sequence longer than n_ctx")
        mask = attn_decoder_mask(T, self.dtype)
        x = self._table("token_embedding", self.d.n_vocab)[tokens] + self._table("position_embedding", self.d.n_ctx)[:T][None]
        for i in range(self.d.n_layer):
            x = self.block(f"{self.root}/blocks/{i}", x, mask)
        return O.layer_norm(x, *self.P.norm(f"{self.root}/layer_norm", self.d.n_state))
