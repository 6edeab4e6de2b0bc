"""CPU restatement of the OCP Microscaling (MX) FP8 format used by precision = 2 (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY (tests/, tools/): the product path never imports this.

The reference (Gadersd/stable-diffusion-burn) has no reduced-precision path at all -- its arithmetic is f32 Burn tensors
(src/bin/sample/main.rs:59-64) -- so there is nothing upstream to pin these functions to; they restate the published
OCP MX v1.0 rules the hardware instruction (v_mfma_scale_f32_16x16x128_f8f6f4) implements:
  * element type e4m3 (OCP FP8 "e4m3fn": bias 7, max 448, no infinity, subnormal step 2^-9), round to nearest even,
    saturating at +-448;
  * one shared power-of-two scale (E8M0, value 2^(byte - 127)) per block of 32 consecutive elements along the
    contraction axis: 2^(floor(log2(amax)) - 8), 8 = emax of e4m3.
`conv_res_mx` is the ResBlock convolution of unet/mod.rs:713-733 with its GroupNorm+SiLU output and its weight in MXFP8 and
fp64 accumulation: what csrc/k_fp8.hip computes up to fp32 accumulation order and the bf16 rounding of the output.
"""
from __future__ import annotations

import numpy as np
import torch


def e4m3_round(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even onto the e4m3 grid, saturating at 448"""
    a = x.abs().clamp(max=448.0)
    e = torch.floor(torch.log2(a.clamp(min=2.0 ** -9))).clamp(min=-6.0)
    step = torch.pow(2.0, e - 3)
    q = torch.round(a / step) * step          # torch.round is round-half-to-even
    return torch.sign(x) * q.clamp(max=448.0)


def mx_scale_exponent(amax: torch.Tensor) -> torch.Tensor:
    """floor(log2(amax)) - 8, with the E8M0 byte kept in [1, 253] like the kernels do (amax = 0 -> tiny scale, all zeros)"""
    e = torch.floor(torch.log2(amax.clamp(min=2.0 ** -140))) - 8
    return e.clamp(min=1 - 127, max=253 - 127)


def mx_quantize(x: torch.Tensor, axis: int) -> torch.Tensor:
    """x with blocks of 32 along `axis` replaced by their MXFP8 values (dequantised, same dtype)."""
    x = x.movedim(axis, -1)
    shp = x.shape
    k = shp[-1]
    pad = (-k) % 32
    xp = torch.nn.functional.pad(x, (0, pad)).reshape(*shp[:-1], -1, 32)
    amax = xp.abs().amax(dim=-1, keepdim=True)
    scale = torch.pow(2.0, mx_scale_exponent(amax)).to(x.dtype)
    q = e4m3_round(xp / scale) * scale
    return q.reshape(*shp[:-1], -1)[..., :k].movedim(-1, axis)


def conv_res_mx(x_gn: torch.Tensor, w: torch.Tensor, b, padding: int = 1) -> torch.Tensor:
    """3x3 conv with MX-quantised input (blocks along channels, per pixel) and weight (blocks along input channels, per tap)."""
    xq = mx_quantize(x_gn, 1)
    wq = mx_quantize(w, 1)
    return torch.nn.functional.conv2d(xq, wq, b, padding=padding)


# ---- the oracle network with the quantisation precision = 2 applies (tests/test_fp8_gpu.py, tests/golden/gen_golden_cfg5.py) ------------
from oracle import sd_oracle as _O  # noqa: E402


class MxResConvs:
    """context manager: the oracle's ResBlock / ResnetBlock 3x3 convs take MXFP8 inputs and weights (what precision = 2 does)"""

    def __enter__(self):
        self.conv0 = _O.conv2d
        self.res0 = _O.UNetOracle.res_block
        state = {"in_res": 0}

        def conv(x, wb, stride=1, padding=0):
            w, b = wb
            if state["in_res"] and w.shape[2] == 3 and stride == 1 and w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0:
                return self.conv0(mx_quantize(x, 1), (mx_quantize(w, 1), b), stride, padding)
            return self.conv0(x, wb, stride, padding)

        def res_block(obj, *a, **k):
            state["in_res"] += 1
            try:
                return self.res0(obj, *a, **k)
            finally:
                state["in_res"] -= 1

        self.vres0 = _O.DecoderOracle.resnet_block

        def resnet_block(obj, *a, **k):      # the VAE's ResnetBlock (autoencoder/mod.rs:514-527): the same two 3x3 convolutions
            state["in_res"] += 1
            try:
                return self.vres0(obj, *a, **k)
            finally:
                state["in_res"] -= 1

        _O.conv2d = conv
        _O.UNetOracle.res_block = res_block
        _O.DecoderOracle.resnet_block = resnet_block
        return self

    def __exit__(self, *exc):
        _O.conv2d = self.conv0
        _O.UNetOracle.res_block = self.res0
        _O.DecoderOracle.resnet_block = self.vres0
