"""Numerical study for the next fp32-GEMM form (CPU, numpy; nothing here runs on the GPU path): a TWO-term fp16 split of each fp32 operand,
x = h + l with h = fp16(s x), l = fp16(s x - h) and a power-of-two scale s, against the shipped THREE-term bf16 split with six partial products
(csrc/k_split3.hpp, oracle/split_oracle.py).  A bf16 x bf16 and an fp16 x fp16 product are both exact in fp32 and the matrix instruction
accumulates in fp32, so what a form loses is (a) what its planes do not represent and (b) the partial products it drops; the fp32 summation
error is common to every form and to the fp32 matrix instruction.  Isolated here by summing the partial products in fp64.

    six bf16 products : h m l of 8 + 8 + 8 bits, products wl ah, wh al, wm am, wm ah, wh am, wh ah            -> 6 matrix instructions per block
    fp16 x 2, 3 prods : h l of 11 + 11 bits, products wh ah, wh al, wl ah                                       -> 3
    fp16 x 2, 4 prods : + wl al                                                                                  -> 4

Prints max|C_form - C_exact| / max|C_exact| for GEMM shapes of the batch-1 UNet, three activation distributions, and -- the catch -- the
dependence on where the activations sit in fp16's range when NO scale is applied (fp16: max 65504, l is subnormal below |x| = 2^-3).
"""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def split_bf16_3(x):
    x = x.astype(np.float32)
    h = bf16(x)
    r = (x.astype(np.float64) - h).astype(np.float32)
    m = bf16(r)
    return h, m, bf16((r.astype(np.float64) - m).astype(np.float32))


def split_fp16_2(x, scale):
    with np.errstate(over="ignore"):
        xs = (x * scale).astype(np.float32)
        h = xs.astype(np.float16).astype(np.float64)
        l = (xs.astype(np.float64) - h).astype(np.float16).astype(np.float64)
    return h / scale, l / scale


def pow2_scale(a, axis=None):
    return 2.0 ** (14 - np.ceil(np.log2(np.abs(a).max(axis=axis, keepdims=axis is not None))))


def forms(A, W, sa):
    C = A.astype(np.float64) @ W.astype(np.float64).T
    cmax = np.abs(C).max()
    ah, am, al = split_bf16_3(A)
    wh, wm, wl = split_bf16_3(W)
    c6 = ah @ wh.T + ah @ wm.T + am @ wh.T + am @ wm.T + ah @ wl.T + al @ wh.T
    a1, a2 = split_fp16_2(A, sa)
    w1, w2 = split_fp16_2(W, pow2_scale(W, axis=1))          # weights: one scale per output channel, chosen at load
    c3 = a1 @ w1.T + a1 @ w2.T + a2 @ w1.T
    c4 = c3 + a2 @ w2.T
    c32 = (A.astype(np.float32) @ W.astype(np.float32).T).astype(np.float64)
    return [np.abs(c - C).max() / cmax for c in (c6, c3, c4, c32)]


print("1. split error alone (partial products summed in fp64); last column: a plain fp32 GEMM of this host (its own summation error), for scale")
print(f"{'activations':28s} {'M x N x K':>18s}   six bf16   fp16x2/3   fp16x2/4   plain fp32")
for K, N in ((320, 320), (2880, 320), (5760, 640), (11520, 1280), (23040, 1280)):
    M = 256
    for name, gen in (("normal", lambda: rng.standard_normal((M, K))),
                      ("ten decades per channel", lambda: rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-5, 5, (1, K))),
                      ("SiLU-like, heavy tail", lambda: np.maximum(rng.standard_t(3, (M, K)), -0.2))):
        A = gen().astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        e = forms(A, W, pow2_scale(A))
        print(f"{name:28s} {M:5d} x{N:5d} x{K:6d}   {e[0]:.2e}   {e[1]:.2e}   {e[2]:.2e}   {e[3]:.2e}")

print()
print("2. fp16 x 2, three products, activations ~ N(0, sigma^2), K = 2880: with a per-tensor power-of-two scale / with NO scale (s = 1)")
M, N, K = 256, 320, 2880
W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
for sigma in (1e-6, 1e-4, 1e-2, 1.0, 1e2, 1e4, 1e5):
    A = (rng.standard_normal((M, K)) * sigma).astype(np.float32)
    e_s = forms(A, W, pow2_scale(A))[1]
    e_1 = forms(A, W, 1.0)[1]
    print(f"   sigma = {sigma:7.0e}: scaled {e_s:.2e}   unscaled {e_1:.2e}")
