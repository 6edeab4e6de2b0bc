"""Accuracy budget for reduced-precision GEMM inputs, measured on the CPU oracle (planning aid for BASELINE configs[4], fp8).

    python tools/quant_study.py [--full]

Every conv / linear of the oracle UNet gets its INPUT and WEIGHT quantised (fp32 accumulation, as the matrix cores do):
  bf16      round-to-nearest-even to bfloat16                       -- calibration: compare with the measured GPU bf16 path
  fp8       OCP e4m3 with one scale per tensor (amax -> 448)
  mxfp8     OCP MX: e4m3 elements, one power-of-two (E8M0) scale per 32 consecutive channels (the K axis of the GEMMs)
and the relative RMS error of one UNet forward against the fp64 oracle is printed.  Default: the half-width test model
(seconds); --full: SD v1.4 size (about a minute per mode).
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import sd_oracle as O  # noqa: E402
from stable_diffusion_burn_amd import synthetic as syn  # noqa: E402


def to_bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


def e4m3_round(x):
    """round-to-nearest-even onto the OCP e4m3 grid (max 448, min normal 2^-6, subnormal step 2^-9), saturating"""
    a = x.abs().clamp(max=448.0)
    e = torch.floor(torch.log2(a.clamp(min=2.0 ** -9)))
    e = e.clamp(min=-6.0)
    step = torch.pow(2.0, e - 3)
    q = torch.round(a / step) * step
    return torch.sign(x) * q.clamp(max=448.0)


def fp8_tensor(x):
    s = x.abs().max().clamp(min=1e-30) / 448.0
    return e4m3_round(x / s) * s


def mxfp8(x, axis):
    """blocks of 32 along `axis`, shared scale 2^(floor(log2(amax)) - 8) (e4m3 emax = 8)"""
    x = x.movedim(axis, -1)
    shp = x.shape
    k = shp[-1]
    pad = (-k) % 32
    xp = torch.nn.functional.pad(x, (0, pad)).reshape(*shp[:-1], -1, 32)
    amax = xp.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -120)
    scale = torch.pow(2.0, torch.floor(torch.log2(amax)) - 8)
    q = e4m3_round(xp / scale) * scale
    return q.reshape(*shp[:-1], -1)[..., :k].movedim(-1, axis)


MODES = {
    "bf16": (lambda x, ax: to_bf16(x), lambda w, ax: to_bf16(w)),
    "fp8": (lambda x, ax: fp8_tensor(x), lambda w, ax: fp8_tensor(w)),
    "mxfp8": (lambda x, ax: mxfp8(x, ax), lambda w, ax: mxfp8(w, ax)),
}


def run(dims, mode):
    qa, qw = MODES[mode]
    conv0, lin0 = O.conv2d, O.linear

    def conv_q(x, wb, stride=1, padding=0):
        w, b = wb
        if w.shape[1] < 32:           # the Cin = 4 layers stay fp32 in every precision
            return conv0(x, wb, stride, padding)
        return conv0(qa(x, 1), (qw(w, 1), b), stride, padding)      # K axis = input channels

    def lin_q(x, w, b):
        return lin0(qa(x, -1), qw(w, 0), b)                         # W is [in, out]

    lat = torch.from_numpy(syn.initial_latent(0, dims.latent_h, dims.latent_w))[None]
    ctx = torch.from_numpy(syn.cond_context(0, 77, dims.ctx_dim))[None]
    ref = O.UNetOracle(syn.SyntheticWeights(), dims, torch.float64).forward(lat, 999, ctx)
    O.conv2d, O.linear = conv_q, lin_q
    try:
        got = O.UNetOracle(syn.SyntheticWeights(), dims, torch.float32).forward(lat, 999, ctx)
    finally:
        O.conv2d, O.linear = conv0, lin0
    d = (got.double() - ref)
    return float(torch.sqrt((d * d).mean()) / torch.sqrt((ref * ref).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    args = ap.parse_args()
    dims = O.Dims() if args.full else O.Dims(160, 4, 64, 16, 16, 32)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for mode in MODES:
        print(f"{mode:6s} GEMM inputs: relative RMS of one UNet forward vs fp64 = {run(dims, mode):.3e}", flush=True)


if __name__ == "__main__":
    main()
