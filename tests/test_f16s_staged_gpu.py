"""STAGED -- skipped unless SDMI_STAGED=1.  The two-term fp16 form of the fp32 GEMM (csrc/k_split2h.hip, k_gemm3p.hip NPL = 2, tile_cfg 400 + x;
DESIGN.md section 10) was written after round 3's GPU minutes were spent and has NEVER run on a GPU: these are the tests its first GPU minute
runs (tools/probes/r04a_session.sh).  Nothing on a default path reaches the code under test (option gemm_f16s, default 0).

Reference arithmetic: Burn's Conv2d / Linear (unet/mod.rs:397,425,468,479,553,580,645-651,716,726,729) in fp32; checker: the fp64 oracle, at the
bar every fp32 operator holds (tests/test_ops_gpu.py: 2e-5 max(1, |ref|)) -- the form's own error is <= 2e-7 of max|C| (tests/test_split_oracle_cpu.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import sd_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.staged]

TILES = [400, 401, 402, 403, 404, 405, 406, 407, 408]
CASES = [
    # (n, cin, h, w, cout, k, stride, ups)
    (2, 128, 23, 19, 320, 3, 1, 0), (1, 64, 40, 36, 200, 3, 1, 0), (2, 192, 16, 16, 640, 1, 1, 0), (1, 128, 33, 31, 128, 3, 2, 0),
    (1, 64, 12, 20, 384, 3, 1, 1), (1, 96, 9, 7, 100, 3, 1, 0), (2, 320, 16, 16, 320, 3, 1, 0),
]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


class _F16s:
    def __init__(self, sd, tile, splitk=0):
        self.sd, self.tile, self.splitk = sd, tile, splitk

    def __enter__(self):
        self.sd.set_option("gemm_f16s", 1)
        self.sd.set_option("gemm_tile", self.tile)
        self.sd.set_option("splitk", self.splitk)
        return self.sd

    def __exit__(self, *a):
        self.sd.set_option("gemm_tile", "auto")
        self.sd.set_option("splitk", 0)
        self.sd.set_option("gemm_f16s", 0)


# every tile on three of the cases (rotating), split-K on the three tiles the batch-1 model would use most
MATRIX = [(t, 1, CASES[(i + j) % len(CASES)]) for i, t in enumerate(TILES) for j in (0, 2, 4)] + [(t, 3, c) for t in (400, 403, 404) for c in (CASES[0], CASES[6])]


@pytest.mark.parametrize("tile,splitk,case", MATRIX)
def test_conv2d_two_term_fp16(sd_ops, tile, splitk, case):
    n, cin, h, w, cout, k, stride, ups = case
    g = np.random.default_rng(4000 + tile + 7 * splitk + cin + cout)
    x = (g.standard_normal((n, cin, h, w)) * 10.0 ** g.uniform(-2, 2, (1, cin, 1, 1))).astype(np.float32)     # four decades across channels
    wt = (g.standard_normal((cout, cin, k, k)) / math.sqrt(cin * k * k) * 10.0 ** g.uniform(-1, 1, (cout, 1, 1, 1))).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    with _F16s(sd_ops, tile, splitk):
        got = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
        again = sd_ops.op_conv2d(x, wt, b, stride=stride, upsample2x=bool(ups))
    xin = O.upsample2x(_t(x)) if ups else _t(x)
    ref = O.conv2d(xin, (_t(wt), _t(b)), stride=stride, padding=1 if k == 3 else 0).numpy()
    assert np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), f"two-term fp16 conv tile={tile} splitk={splitk} {case}: {err:.3e}"
    assert np.array_equal(got, again)


@pytest.mark.parametrize("rows,cin,cout", [(154, 768, 320), (20, 320, 1280), (512, 320, 2560), (1, 1280, 1280), (77, 64, 160)])
def test_linear_two_term_fp16(sd_ops, rows, cin, cout):
    g = np.random.default_rng(rows + cin)
    x = g.standard_normal((rows, cin)).astype(np.float32)
    w = (g.standard_normal((cin, cout)) / math.sqrt(cin)).astype(np.float32)
    b = g.standard_normal(cout).astype(np.float32)
    with _F16s(sd_ops, 404):
        got = sd_ops.op_linear(x, w, b)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_two_term_fp16_is_as_accurate_as_the_six_product_form_needs_to_be(sd_ops):
    """K = 11520, inputs spanning ten binary orders per channel: the staged form against the fp64 oracle next to the shipped plane kernel
    (tests/test_planes_gpu.py measured 2.8e-6 for the six-product form, 4.6e-6 for the fp32 matrix instruction on this case)."""
    g = np.random.default_rng(77)
    n, cin, h, w, cout = 1, 1280, 16, 16, 320
    x = (g.standard_normal((n, cin, h, w)) * np.exp2(g.integers(-5, 5, (1, cin, 1, 1)))).astype(np.float32)
    wt = (g.standard_normal((cout, cin, 3, 3)) / math.sqrt(cin * 9)).astype(np.float32)
    ref = O.conv2d(_t(x), (_t(wt), None), stride=1, padding=1).numpy()
    with _F16s(sd_ops, 400):
        got = sd_ops.op_conv2d(x, wt, None)
    base = sd_ops.op_conv2d(x, wt, None)
    e2 = np.abs(got - ref).max() / np.abs(ref).max()
    e6 = np.abs(base - ref).max() / np.abs(ref).max()
    print(f"K = 11520: two-term fp16 {e2:.2e}, six-product bf16 {e6:.2e} of max|ref|")
    assert e2 < 1e-5
