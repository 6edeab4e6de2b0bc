"""The MX (OCP microscaling) FP8 restatement used as the checker of the precision = 2 path: properties of the format
itself, on the CPU (the kernels are compared with it in tests/test_fp8_gpu.py)."""
import numpy as np
import torch

from oracle import mx_oracle as MX


def test_e4m3_grid_and_saturation():
    x = torch.tensor([0.0, 1.0, 1.0625, 1.1875, 448.0, 464.0, 480.0, 1000.0, -0.001953125, 0.0009765625, 0.017, 3.3, -449.0, 240.0, 2.5], dtype=torch.float64)
    # the values v_cvt_pk_fp8_f32 returned for the same inputs on MI355X (profiles/r02_mx_mfma_cvt_probe.txt), with the
    # kernels' clamp to 448 applied where the instruction alone gives NaN (480, 1000)
    want = torch.tensor([0.0, 1.0, 1.0, 1.25, 448.0, 448.0, 448.0, 448.0, -0.001953125, 0.0, 0.017578125, 3.25, -448.0, 240.0, 2.5], dtype=torch.float64)
    assert torch.equal(MX.e4m3_round(x), want)


def test_every_e4m3_code_is_a_fixed_point():
    codes = np.arange(256, dtype=np.uint8)
    e, m = (codes >> 3) & 15, codes & 7
    val = np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    val = np.where(codes & 0x80, -val, val)
    ok = (codes & 0x7f) != 0x7f
    v = torch.from_numpy(val[ok])
    assert torch.equal(MX.e4m3_round(v), v)


def test_block_scale_rule_and_idempotence():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 96, 5, 5, generator=g, dtype=torch.float64) * torch.logspace(-3, 3, 96, dtype=torch.float64).reshape(1, 96, 1, 1)
    q = MX.mx_quantize(x, 1)
    assert torch.equal(MX.mx_quantize(q, 1), q)                    # values on the grid stay where they are
    blocks = x.movedim(1, -1).reshape(4, 5, 5, 3, 32)
    qb = q.movedim(1, -1).reshape(4, 5, 5, 3, 32)
    amax = blocks.abs().amax(-1, keepdim=True)
    scale = 2.0 ** (torch.floor(torch.log2(amax)) - 8)
    assert (qb.abs().amax(-1, keepdim=True) / scale <= 448).all()
    rel = ((qb - blocks).abs() / amax).max()
    assert rel <= 64 / 512 + 1e-12                                   # worst case: the block maximum clamped from <512 to 448
    # elements not touched by the clamp are within half an e4m3 step of their block-scaled value
    scaled = blocks / scale
    inside = scaled.abs() <= 448
    step = 2.0 ** (torch.floor(torch.log2(scaled.abs().clamp(min=2.0 ** -9))).clamp(min=-6) - 3)
    assert ((qb / scale - scaled).abs()[inside] <= 0.5 * step[inside] + 1e-12).all()
