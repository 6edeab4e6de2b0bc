"""The data path of the plane GEMM (k_gemm3p.hip) on the host: HBM planes -> per-wave DMA pieces with the slot swizzle -> LDS bytes -> per-lane
fragment reads -> the 16x16x32 matrix instruction's operand / accumulator layout -> epilogue, for one launch of a small 3x3 convolution and of a
Linear layer (three bf16 planes, six products).
Every index formula is transcribed from the kernel (piece<J>(), the prologue's a_off / w_off / fr / a_fr / w_fr, read_a / read_w, the
epilogue's lane -> (pixel, channel) map); what the test adds to reading them is that they are exercised together, with padding taps, a ragged
last M tile, a ragged last N tile and more weight pieces than waves."""
import numpy as np
import pytest

from oracle import split_oracle as S


def plane_pos(j):                                   # k_split3.hpp s3_plane_pos
    return ((j & 15) >> 2) * 8 + ((j >> 4) << 2) + (j & 3)


POS = np.array([plane_pos(j) for j in range(32)])


def to_planes(x, npl):
    """x [rows, C] fp32 -> [rows, C / 32, npl, 32] float64 plane values in the kernels' order (k_split3.hpp)"""
    rows, c = x.shape
    terms = S.split3(x)
    out = np.zeros((rows, c // 32, npl, 32))
    for pl, t in enumerate(terms):
        t = t.reshape(rows, c // 32, 32).astype(np.float64)
        out[:, :, pl, POS] = t
    return out


PRODUCTS = {3: ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))}     # (weight plane, activation plane), k_gemm3p.hip mfmas()


def emulate(npl, MI, NI, WM, WN, A3, W3, nb, hs, ws, cin, N, k, stride):
    """one launch, no split-K: returns C [M, N] (accumulated in float64 in the kernel's product order)"""
    pad = 1 if k == 3 else 0
    ho, wo = (hs + 2 * pad - k) // stride + 1, (ws + 2 * pad - k) // stride + 1
    M, T, kt_total = nb * ho * wo, k * k, (cin // 32) * k * k
    BM, BN, NWV = 16 * MI * WM, 16 * NI * WN, WM * WN
    NAG = BM // (16 * NWV)
    PW = (BN // 16) * npl
    NBW = (PW + NWV - 1) // NWV
    A_PIECES = (BM // 16) * npl
    f = lambda r: (-(r >> 2)) & 3
    C = np.zeros((M, N))
    for tm in range((M + BM - 1) // BM):
        for tn in range((N + BN - 1) // BN):
            m0, n0 = tm * BM, tn * BN
            acc = np.zeros((NWV, MI, NI, 16, 16))                       # [wave][fragment row][fragment column][i = channel][j = pixel]
            cs = ky = kx = 0
            for kt in range(kt_total):
                lds = np.full((A_PIECES + NBW * NWV, 16, 4, 8), np.nan)  # [piece][row][slot][8 elements]: a piece is 16 rows x 64 bytes
                for wave in range(NWV):
                    for lane in range(64):
                        r16, slot = lane >> 2, lane & 3
                        ch = slot ^ f(r16)
                        for jg in range(NAG):                           # activation pieces: all planes of the wave's fragment groups
                            G = wave + NWV * jg
                            m = m0 + G * 16 + r16
                            src = np.zeros((npl, 8))
                            if m < M:
                                b, rem = divmod(m, ho * wo)
                                oy, ox = divmod(rem, wo)
                                iy, ix = oy * stride - pad + ky, ox * stride - pad + kx
                                if 0 <= iy < hs and 0 <= ix < ws:
                                    src = A3[(b * hs + iy) * ws + ix, cs, :, ch * 8:ch * 8 + 8]
                            for pl in range(npl):
                                lds[G * npl + pl, r16, slot] = src[pl]
                        for j in range(NBW):                            # weight pieces q = wave + NWV j: plane q % npl of fragment group q / npl
                            q = wave + NWV * j
                            fg, pl = divmod(q, npl)
                            n = n0 + fg * 16 + r16
                            wrow = min(n, N - 1)                        # rows past N re-read the last valid row: real memory, never stored
                            lds[A_PIECES + q, r16, slot] = W3[wrow, kt, pl, ch * 8:ch * 8 + 8]
                for wave in range(NWV):
                    wm, wn = divmod(wave, WN)
                    def frag(piece):                                    # lane (c15, g4) reads row c15, slot g4 ^ f(c15): -> [16 rows][4 chunks][8]
                        return np.stack([np.stack([lds[piece, c, g ^ f(c)] for g in range(4)]) for c in range(16)])
                    af = [[frag((wm * MI + F) * npl + pl) for pl in range(npl)] for F in range(MI)]
                    wf = [[frag(A_PIECES + (wn * NI + n_) * npl + pl) for n_ in range(NI)] for pl in range(npl)]
                    for F in range(MI):
                        for wp, ap in PRODUCTS[npl]:
                            for n_ in range(NI):
                                if (wn * NI + n_) * npl + npl - 1 < PW:  # fragment columns past the tile's weight pieces do not exist
                                    acc[wave, F, n_] += np.einsum("ige,jge->ij", wf[wp][n_], af[F][ap])
                kx += 1                                                  # piece<NP - 1>(): taps inner (kx fastest), channel slices outer
                if kx == k:
                    kx, ky = 0, ky + 1
                    if ky == k:
                        ky, cs = 0, cs + 1
            for wave in range(NWV):                                      # epilogue: lane (c15, g4) holds channels 4 g4 .. 4 g4 + 3 of pixel c15
                wm, wn = divmod(wave, WN)
                for F in range(MI):
                    for n_ in range(NI):
                        for i in range(16):
                            for j in range(16):
                                m, n = m0 + (wm * MI + F) * 16 + j, n0 + (wn * NI + n_) * 16 + i
                                if m < M and n < N:
                                    C[m, n] = acc[wave, F, n_, i, j]
    assert not np.isnan(C).any()
    return C


def reference(x_nhwc, w_oihw, k, stride):
    nb, hs, ws, cin = x_nhwc.shape
    pad = 1 if k == 3 else 0
    ho, wo = (hs + 2 * pad - k) // stride + 1, (ws + 2 * pad - k) // stride + 1
    xp = np.zeros((nb, hs + 2 * pad, ws + 2 * pad, cin))
    xp[:, pad:pad + hs, pad:pad + ws] = x_nhwc
    out = np.zeros((nb, ho, wo, w_oihw.shape[0]))
    for ky in range(k):
        for kx in range(k):
            patch = xp[:, ky:ky + (ho - 1) * stride + 1:stride, kx:kx + (wo - 1) * stride + 1:stride]
            out += np.einsum("bhwc,oc->bhwo", patch, w_oihw[:, :, ky, kx])
    return out.reshape(-1, w_oihw.shape[0])


def packed_weight(w_oihw):
    """[cout][K] in the kernels' k order: k = (channel slice * T + tap) * 32 + channel in slice (kernels.hpp)"""
    cout, cin, k, _ = w_oihw.shape
    return w_oihw.reshape(cout, cin // 32, 32, k * k).transpose(0, 1, 3, 2).reshape(cout, -1)


@pytest.mark.parametrize("npl", [3])
@pytest.mark.parametrize("tile", [(2, 2, 2, 2), (2, 4, 4, 2)])          # 64 x 64 with four waves, 128 x 128 with eight
@pytest.mark.parametrize("k,stride,shape", [(3, 1, (1, 9, 8, 64, 80)), (1, 1, (1, 1, 150, 96, 144)), (3, 2, (2, 7, 6, 32, 40))])
def test_one_launch_reproduces_a_w_transposed(npl, tile, k, stride, shape):
    MI, NI, WM, WN = tile
    nb, hs, ws, cin, N = shape
    g = np.random.default_rng(nb + hs + cin + N + npl)
    x = g.standard_normal((nb, hs, ws, cin)).astype(np.float32)
    w = (g.standard_normal((N, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bt = packed_weight(w)
    A3 = to_planes(x.reshape(-1, cin), npl)
    W3 = to_planes(bt, npl)
    got = emulate(npl, MI, NI, WM, WN, A3, W3, nb, hs, ws, cin, N, k, stride)
    # what the kept partial products of the represented operands add up to, independent of the kernel's indexing
    a_terms = [A3[:, :, pl][:, :, POS].reshape(nb, hs, ws, cin) for pl in range(npl)]
    w_terms = [W3[:, :, pl][:, :, POS].reshape(N, cin // 32, k * k, 32).transpose(0, 1, 3, 2).reshape(N, cin, k, k) for pl in range(npl)]
    want = sum(reference(a_terms[ap], w_terms[wp], k, stride) for wp, ap in PRODUCTS[npl])
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    exact = reference(x.astype(np.float64), w.astype(np.float64), k, stride)
    assert np.abs(got - exact).max() <= 1e-7 * np.abs(exact).max()
