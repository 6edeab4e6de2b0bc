"""The arithmetic claims behind precision = 0's split kernels, as CPU theorems over the oracle restatement
(oracle/split_oracle.py).  The GPU side of the same claims is tests/test_ops_gpu.py (test_conv2d_split_bf16_*)."""
import numpy as np

from oracle import split_oracle as S


def _samples():
    g = np.random.default_rng(2024)
    x = g.standard_normal(200000).astype(np.float32)
    wide = (g.standard_normal(200000) * np.exp2(g.integers(-60, 60, 200000))).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 2.0 ** -100, 1.17549435e-38 * 2 ** 20,
                     255.0, 256.0, 257.0, 65535.0, 16777215.0, 0.1, 1.0 / 3.0, np.pi], np.float32)
    ties = (np.arange(1, 4097, dtype=np.float32) * np.float32(2.0 ** -7) + np.float32(1.0))      # many exact bf16 rounding ties
    return np.concatenate([x, wide, edge, ties])


def test_bf16_rne_matches_torch():
    import torch
    x = _samples()
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(S.bf16_rne(x), want)


def test_three_terms_are_bf16_and_sum_exactly():
    x = _samples()
    h, m, l = S.split3(x)
    for t in (h, m, l):
        assert np.array_equal(S.bf16_rne(t), t)                       # each term is representable in bf16
    s = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))                    # exact: 8 + 8 + 8 significand bits cover fp32's 24
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all()        # |m| <= ulp_bf16(x) / 2 <= 2^-8 |x| (2^-9 away from power-of-two edges)
    assert (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_the_three_dropped_products_are_below_half_an_fp32_rounding():
    g = np.random.default_rng(7)
    a = (g.standard_normal(300000) * np.exp2(g.integers(-20, 20, 300000))).astype(np.float32)
    w = (g.standard_normal(300000) * np.exp2(g.integers(-20, 20, 300000))).astype(np.float32)
    kept, dropped = S.six_products(a, w)
    exact = a.astype(np.float64) * w.astype(np.float64)
    assert np.allclose(sum(kept) + sum(dropped), exact, rtol=0, atol=0)          # nine products = the exact product (all exact in float64)
    err = np.abs(sum(dropped))
    nz = exact != 0
    rel = err[nz] / np.abs(exact[nz])
    print(f"dropped terms / |a w|: max {rel.max():.3e} (2^-24 = {2.0 ** -24:.3e}), mean {rel.mean():.3e}")
    assert rel.max() <= 2.0 ** -23.4          # worst case (1 + 2^-8)^2 * (2^-25 + 2^-25 + 2^-33): below one fp32 rounding of the product (2^-24 relative to the next power of two)
    assert rel.mean() <= 2.0 ** -27
    for t in kept:                                                                # every kept partial product is exact in fp32 (8 x 8 bits)
        assert np.array_equal(t.astype(np.float32).astype(np.float64), t)


def test_split_gemm_is_as_accurate_as_an_fp32_gemm():
    g = np.random.default_rng(11)
    M, N, K = 48, 40, 2304
    a = (g.standard_normal((M, K)) * np.exp2(g.integers(-5, 6, (1, K)))).astype(np.float32)
    w = (g.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    split = S.gemm_split(a, w)                       # the split's own error, summation in float64
    f32 = (a @ w.T).astype(np.float64)               # a plain fp32 GEMM (BLAS): rounds every partial sum
    scale = np.abs(exact).max()
    e_split, e_f32 = np.abs(split - exact).max() / scale, np.abs(f32 - exact).max() / scale
    print(f"K = {K}: split (six products) {e_split:.2e}, plain fp32 GEMM {e_f32:.2e} of max|ref|")
    assert e_split < 2.0 ** -24
    assert e_split < e_f32                            # what the split drops is less than what fp32 summation rounds away


def test_small_integers_need_the_low_planes_and_come_out_exact():
    g = np.random.default_rng(5)
    a = g.integers(-300, 301, (16, 576)).astype(np.float32)
    w = g.integers(-40, 41, (24, 576)).astype(np.float32)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(exact).max() < 2 ** 24
    assert np.array_equal(S.gemm_split(a, w), exact)
    h = S.split3(a)[0]
    assert not np.array_equal(h, a)                   # bf16 alone (the h plane) cannot hold 9-bit integers


def test_reduced_precision_geglu_gate_formula():
    """The GELU of the GEGLU gate in the bf16 / MXFP8 kernels (csrc/k_common.hpp gelu_gate_fast; reference unet/mod.rs:587-590: x * 0.5 * (1 + erf(x / sqrt 2))):
    Phi(g) = 0.5 erfc(-g / sqrt 2) with Abramowitz-Stegun 7.1.26 on a reciprocal and an exp2, restated here in float32 step by step.  Its distance from the exact gate is
    below 6e-7 over |g| <= 12 -- three orders of magnitude inside the bf16 rounding of the result -- and a result lands on a different bf16 value than the exact gate's in
    under 2 % of normally distributed inputs (then on the adjacent one)."""
    from scipy.special import erf
    f = np.float32

    def gate(g):
        g = g.astype(f)
        x = (np.abs(g) * f(0.70710678118654752440)).astype(f)
        t = (f(1.0) / (f(0.3275911) * x + f(1.0)).astype(f)).astype(f)
        a = [f(0.5 * v) for v in (0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429)]
        p = (t * a[4] + a[3]).astype(f)
        for k in (2, 1, 0):
            p = (p * t + a[k]).astype(f)
        h = ((p * t).astype(f) * np.exp2(((g * g).astype(f) * f(-0.72134752044448170368)).astype(f)).astype(f)).astype(f)
        return (g * (f(0.5) + np.copysign((f(0.5) - h).astype(f), g)).astype(f)).astype(f)

    def exact(g):
        g = g.astype(np.float64)
        return g * 0.5 * (1.0 + erf(g / np.sqrt(2.0)))

    g = np.linspace(-12.0, 12.0, 1200001).astype(f)
    assert np.abs(gate(g).astype(np.float64) - exact(g)).max() < 6e-7
    assert gate(np.array([0.0, -0.0], f)).tolist() == [0.0, 0.0] and np.isfinite(gate(np.array([-80.0, 80.0, 1e-30], f))).all()
    assert gate(np.array([80.0], f))[0] == f(80.0) and abs(float(gate(np.array([-80.0], f))[0])) == 0.0

    def bf16(x):
        u = x.astype(f).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(f)

    gg = (np.random.default_rng(0).standard_normal(400000) * 2).astype(f)
    a, b = bf16(gate(gg)), bf16(exact(gg).astype(f))
    assert np.mean(a != b) < 0.02
    assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= np.abs(b).max() * 2.0 ** -7
