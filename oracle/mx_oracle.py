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
    """context manager: the oracle network with the MXFP8 quantisation precision = 2 applies.

    wide = False (option fp8_linear = 0, round 2): the ResBlock / ResnetBlock 3x3 convolutions take MXFP8 inputs and weights.
    wide = True  (the default of precision = 2 since round 3): additionally every Linear layer of the transformer blocks (q | k | v of the
    self-attention, the attention out-projections, the cross-attention query, the GEGLU projection, the MLP's second Linear -- not the
    cross-attention K / V of the text context, which the engine hoists out of the step loop in bf16), the SpatialTransformer's 1x1
    proj_in / proj_out, the ResBlocks' 1x1 shortcut convolutions and the UNet's down / up convolutions.  Attention itself, the Cin = 4 /
    Cout <= 4 layers and the time-embedding MLP stay unquantised, as on the GPU; so does everything of the VAE decoder except its ResnetBlock
    3x3 convolutions in BOTH modes (measured on MI355X, round 3: with the decoder's up-convolutions and 1x1 shortcuts in MXFP8 too the decoded RGB
    sits 9.8e-2 relative RMS from the exact decode instead of 2.1e-2, for 0.4 ms per image)."""

    def __init__(self, wide: bool = False):
        self.wide = wide

    def __enter__(self):
        self.conv0 = _O.conv2d
        self.lin0 = _O.linear
        self.res0 = _O.UNetOracle.res_block
        self.vres0 = _O.DecoderOracle.resnet_block
        self.st0 = _O.UNetOracle.spatial_transformer
        self.vattn0 = _O.DecoderOracle.attn_block
        self.vdec0 = _O.DecoderOracle.decode_latent
        state = {"in_res": 0, "st_c": None, "no_q": 0, "in_vae": 0}
        wide = self.wide

        def conv(x, wb, stride=1, padding=0):
            w, b = wb
            ok_shape = w.shape[1] % 32 == 0 and w.shape[0] % 8 == 0 and not state["no_q"]
            res3 = state["in_res"] and w.shape[2] == 3 and stride == 1
            if ok_shape and (res3 or (wide and not state["in_vae"] and w.shape[2] in (1, 3))):
                return self.conv0(mx_quantize(x, 1), (mx_quantize(w, 1), b), stride, padding)
            return self.conv0(x, wb, stride, padding)

        def linear(x, w, b):
            c = state["st_c"]
            if wide and c is not None and w.shape[0] in (c, 4 * c) and w.shape[0] % 32 == 0 and w.shape[1] % 8 == 0:
                return self.lin0(mx_quantize(x, -1), mx_quantize(w, 0), b)        # w is [in, out]: blocks along the contraction axis
            return self.lin0(x, w, b)

        def counted(key, fn):
            def wrapped(obj, *a, **k):
                state[key] += 1
                try:
                    return fn(obj, *a, **k)
                finally:
                    state[key] -= 1
            return wrapped

        def spatial(obj, path, x, context, c):
            prev = state["st_c"]
            state["st_c"] = c
            try:
                return self.st0(obj, path, x, context, c)
            finally:
                state["st_c"] = prev

        _O.conv2d = conv
        _O.linear = linear
        _O.UNetOracle.res_block = counted("in_res", self.res0)
        _O.DecoderOracle.resnet_block = counted("in_res", self.vres0)     # the VAE's ResnetBlock (autoencoder/mod.rs:514-527): the same two 3x3 convolutions
        _O.DecoderOracle.attn_block = counted("no_q", self.vattn0)
        _O.DecoderOracle.decode_latent = counted("in_vae", self.vdec0)
        _O.UNetOracle.spatial_transformer = spatial
        return self

    def __exit__(self, *exc):
        _O.conv2d = self.conv0
        _O.linear = self.lin0
        _O.UNetOracle.res_block = self.res0
        _O.DecoderOracle.resnet_block = self.vres0
        _O.DecoderOracle.attn_block = self.vattn0
        _O.DecoderOracle.decode_latent = self.vdec0
        _O.UNetOracle.spatial_transformer = self.st0
