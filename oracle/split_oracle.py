"""Oracle for the three-way bf16 split arithmetic of precision = 0 (csrc/k_gemm3x.hip, csrc/k_attn_split.hip).

TEST INFRASTRUCTURE ONLY (tests/ may import it; the product never does).  The reference (Gadersd/stable-diffusion-burn) computes
in plain fp32 and has no counterpart of this; what is restated here is IEEE arithmetic: bf16 = the upper 16 bits of an fp32 number,
conversion by round-to-nearest-even (what v_cvt_pk_bf16_f32 does on gfx950).  The functions are exact numpy integer / float64
manipulations, so the CPU tests can state the kernels' claims as theorems over random and adversarial inputs:

    split3(x)        -> (h, m, l) bf16-representable fp32 numbers with h + m + l == x exactly
    six_products()   -> the six partial products the kernels accumulate, and the three they drop
"""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32.  Finite inputs below the bf16 overflow threshold."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


def split3(x):
    """x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); the subtractions are exact in fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    r2 = (r - m).astype(np.float32)
    l = bf16_rne(r2)
    return h, m, l


def six_products(a, w):
    """float64 values of the six partial products the kernels accumulate (smallest first) and of the three they drop."""
    ah, am, al = (v.astype(np.float64) for v in split3(a))
    wh, wm, wl = (v.astype(np.float64) for v in split3(w))
    kept = [wl * ah, wh * al, wm * am, wm * ah, wh * am, wh * ah]
    dropped = [wm * al, wl * am, wl * al]
    return kept, dropped


def gemm_split(a, w):
    """a [M, K] @ w [N, K]^T the way the split kernel forms it: six exact partial products per (m, n, k), summed here in float64
    (the kernel sums in fp32; this isolates what the SPLIT costs).  Small shapes only."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    ap = [v.astype(np.float64) for v in split3(a)]
    wp = [v.astype(np.float64) for v in split3(w)]
    out = np.zeros((a.shape[0], w.shape[0]), np.float64)
    for ia, iw in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)):   # (a plane, w plane): h = 0, m = 1, l = 2
        out += ap[ia] @ wp[iw].T
    return out
