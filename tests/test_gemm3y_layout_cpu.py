"""Index arithmetic of k_gemm3y.hip and k_gemm_bf16y.hip (the 32x32x16 GEMM families, not yet run on a GPU), emulated on the CPU.

The kernel's correctness hangs on a chain of layouts that must agree with each other: the DMA's source-side swizzles, the LDS addresses
of the fragment reads, which eight k-values a lane of v_mfma_f32_32x32x16_bf16 supplies on the weight and on the activation side, the
plane packing of launch_pack_split3, and the accumulator layout the epilogue unpacks.  This test restates every formula of the kernel
(line references in the comments) in numpy, moves a random tile through "HBM -> LDS bytes -> lane registers -> matrix instruction ->
accumulators -> output" for every wave of a workgroup, and compares with A @ W^T.  The matrix instruction itself is modelled by its
documented operand layout (lane l: row / column l & 31, k-slots 8 (l >> 5) .. + 7; D: lane holds column l & 31, rows
(r & 3) + 8 (r >> 2) + 4 (l >> 5)) -- the layout k_attn_bf16.hip already relies on on hardware.  It also checks that the fragment reads
are free of LDS bank conflicts under the documented ds_read_b128 lane groups.  Values are kept in fp64 and not split into planes: the
split is value arithmetic (tests/test_split_oracle_cpu.py), the layouts are what is new here; the three planes travel as three copies.
"""
import numpy as np
import pytest

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def plane_chunk_elems(g):
    """launch_pack_split3 (k_gemm3x.hip): chunk g of a plane holds k-tile elements 4g .. 4g+3, 16+4g .. 16+4g+3."""
    return [4 * g + j for j in range(4)] + [16 + 4 * g + j for j in range(4)]


def emulate(ni, wm_n, wn_n, rng):
    bm, bn = 32 * wm_n, 32 * ni * wn_n
    na = bm // 64
    pw = (bn // 16) * 3
    nbw = (pw + 7) // 8
    a_bytes = bm * 128
    stage_elems = (a_bytes + nbw * 8 * 1024) // 4          # LDS modelled as an array of 4-byte cells holding fp64 values (or None)
    A = rng.standard_normal((bm, 32))                      # activations of one k tile: [tile row][k]
    W = rng.standard_normal((bn, 32))                      # weights: [tile column][k]
    # weight planes in HBM: per column 3 planes x 32 elements in chunk order; a 16-byte chunk = 8 bf16 -> modelled as 4 cells of (2 values)
    lds = [None] * stage_elems

    # ---- DMA, activations (kernel lines "activation piece j of a wave"): wave-instruction = 64 lanes x 16 B -> LDS piece base + lane * 16
    for wave in range(8):
        for j in range(na):
            for lane in range(64):
                sub = lane >> 3
                chunk = (lane & 7) ^ ((((wave & 1) << 2) + (sub >> 1)) & 7)
                row = (wave + 8 * j) * 8 + sub
                src = A[row, chunk * 4: chunk * 4 + 4]                    # global chunk `chunk` of the row (16 B = 4 floats)
                dst = ((wave + 8 * j) * 1024 + lane * 16) // 4
                for e in range(4):
                    lds[dst + e] = ("a", row, chunk * 4 + e, src[e])
    # ---- DMA, weight planes: piece q = wave + 8 j: 16-column group q // 3, plane q % 3; lane -> column lane >> 2, chunk (lane & 3) ^ ((-(col >> 2)) & 3)
    for wave in range(8):
        for j in range(nbw):
            q = wave + 8 * j
            f, pl = q // 3, q % 3
            for lane in range(64):
                r = lane >> 2
                ch = (lane & 3) ^ ((-(r >> 2)) & 3)
                col = f * 16 + r
                dst = (a_bytes + (wave + 8 * j) * 1024 + lane * 16) // 4
                if col >= bn:
                    for e in range(4):
                        lds[dst + e] = ("dead",)
                    continue
                elems = plane_chunk_elems(ch)                               # the 8 k-indices this 16-byte chunk of the plane holds, in order
                for e in range(4):                                           # 4 cells of 2 bf16 each
                    lds[dst + e] = ("w", col, pl, (elems[2 * e], elems[2 * e + 1]), (W[col, elems[2 * e]], W[col, elems[2 * e + 1]]))

    out = np.zeros((bm, bn))
    conflicts = 0
    for wave in range(8):
        wm, wn = wave // wn_n, wave % wn_n
        acc = np.zeros((ni, 64, 16))
        for s in range(2):
            a_regs, a_addr = [], ([], [])
            for lane in range(64):
                c, hi = lane & 31, lane >> 5
                arow = wm * 32 + c
                asw = (arow >> 1) & 7
                off0 = arow * 128 + (((2 * s + hi) ^ asw) << 4)              # a_c0[s]
                off1 = arow * 128 + (((4 + 2 * s + hi) ^ asw) << 4)          # a_c1[s]
                a_addr[0].append(off0)
                a_addr[1].append(off1)
                vals, ks = [], []
                for off in (off0, off1):
                    for e in range(4):
                        cell = lds[off // 4 + e]
                        assert cell[0] == "a" and cell[1] == arow, "activation fragment read hits the wrong row"
                        ks.append(cell[2])
                        vals.append(cell[3])
                a_regs.append((ks, vals))                                    # S3SplitT::load order: x0[0..3], x1[0..3] -> bf16x8 element order
            for addrs in a_addr:
                conflicts += bank_conflicts(addrs)
            for f in range(ni):
                for pl in range(3):
                    w_regs, w_addr = [], []
                    for lane in range(64):
                        c, hi = lane & 31, lane >> 5
                        wcol0 = wn * 32 * ni + c
                        w_r16 = wcol0 & 15
                        w_sw = (-(w_r16 >> 2)) & 3
                        off = a_bytes + ((wcol0 >> 4) * 3) * 1024 + w_r16 * 64 + f * 6144 + pl * 1024 + (((2 * s + hi) ^ w_sw) << 4)
                        w_addr.append(off)
                        ks, vals = [], []
                        for e in range(4):
                            cell = lds[off // 4 + e]
                            assert cell[0] == "w" and cell[1] == wcol0 + 32 * f and cell[2] == pl, "weight fragment read hits the wrong column / plane"
                            ks += list(cell[3])
                            vals += list(cell[4])
                        w_regs.append((ks, vals))
                    conflicts += bank_conflicts(w_addr)
                    if pl:
                        continue                                              # the three planes carry the same value here: count the product once
                    # v_mfma_f32_32x32x16_bf16: D[row of A-lane][col of B-lane] += sum over (hi, j) of A(row, hi)[j] * B(col, hi)[j]
                    for lb in range(64):
                        cb, hb = lb & 31, lb >> 5
                        for r in range(16):
                            row = (r & 3) + 8 * (r >> 2) + 4 * hb               # the channel inside the fragment that acc[r] of lane lb holds
                            tot = 0.0
                            for hh in range(2):
                                ka, va = w_regs[row + 32 * hh]                 # A operand: lane (row, hh)
                                kb, vb = a_regs[cb + 32 * hh]                  # B operand: lane (col, hh)
                                assert ka == kb, f"the two operands disagree on the k-values of slot ({s}, {hh}): {ka} vs {kb}"
                                tot += float(np.dot(va, vb))
                            acc[f, lb, r] += tot
        # ---- epilogue (vec_ok path): scratch transpose, then row segments
        for f in range(ni):
            scr = np.full((32, 36), np.nan)
            for lane in range(64):
                c, hi = lane & 31, lane >> 5
                for q in range(4):
                    scr[c, 8 * q + 4 * hi: 8 * q + 4 * hi + 4] = acc[f, lane, 4 * q: 4 * q + 4]
            for it in range(4):
                for lane in range(64):
                    row, c4 = it * 8 + (lane >> 3), lane & 7
                    m, n = wm * 32 + row, (wn * ni + f) * 32 + c4 * 4
                    out[m, n: n + 4] = scr[row, c4 * 4: c4 * 4 + 4]
    return A, W, out, conflicts


def bank_conflicts(byte_addrs):
    """extra LDS cycles of one ds_read_b128 wave-instruction: per 16-lane group, 16-byte accesses on 64 banks of 4 bytes"""
    extra = 0
    for grp in B128_GROUPS:
        slots = {}
        for l in grp:
            slots.setdefault((byte_addrs[l] // 16) % 16, set()).add(byte_addrs[l] // 16)
        extra += max(len(v) for v in slots.values()) - 1
    return extra


@pytest.mark.parametrize("ni,wm,wn", [(5, 4, 2), (5, 8, 1), (4, 4, 2), (4, 8, 1)])
def test_gemm3y_tile_reproduces_a_times_w_transposed(ni, wm, wn):
    rng = np.random.default_rng(100 * ni + wm)
    A, W, out, conflicts = emulate(ni, wm, wn, rng)
    ref = A @ W.T
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12), f"max |diff| = {np.abs(out - ref).max()}"
    assert conflicts == 0, f"{conflicts} extra LDS cycles from bank conflicts in the fragment reads"


def test_every_k_of_a_tile_is_used_exactly_once():
    used = sorted(k for s in range(2) for hi in range(2) for k in plane_chunk_elems(2 * s + hi))
    assert used == list(range(32))


# ---- k_gemm_bf16y.hip -----------------------------------------------------------------------------------------------------------------
def emulate_bf16y(mi_n, ni, wm_n, wn_n, rng, out_f32):
    bm, bn = 32 * mi_n * wm_n, 32 * ni * wn_n
    na, nb = bm // 64, bn // 64
    a_bytes = bm * 128
    A = rng.standard_normal((bm, 64))                      # one k tile: 64 bf16 per row = 8 chunks of 8
    W = rng.standard_normal((bn, 64))
    lds = [None] * (((bm + bn) * 128) // 16)               # 16-byte cells
    for wave in range(8):
        for j in range(max(na, nb)):
            for lane in range(64):
                sub = lane >> 3
                chunk = (lane & 7) ^ ((((wave & 1) << 2) + (sub >> 1)) & 7)
                row = (wave + 8 * j) * 8 + sub
                if j < na:
                    lds[((wave + 8 * j) * 1024 + lane * 16) // 16] = ("a", row, chunk, A[row, chunk * 8: chunk * 8 + 8])
                if j < nb:
                    lds[(a_bytes + (wave + 8 * j) * 1024 + lane * 16) // 16] = ("w", row, chunk, W[row, chunk * 8: chunk * 8 + 8])
    out = np.zeros((bm, bn))
    conflicts = 0
    for wave in range(8):
        wm, wn = wave // wn_n, wave % wn_n
        acc = np.zeros((mi_n, ni, 64, 16))
        for s_ in range(4):
            fa = [[None] * 64 for _ in range(mi_n)]
            fb = [[None] * 64 for _ in range(ni)]
            addr_a = [[0] * 64 for _ in range(mi_n)]
            addr_b = [[0] * 64 for _ in range(ni)]
            for lane in range(64):
                c, hi = lane & 31, lane >> 5
                sw = (c >> 1) & 7
                g_off = ((2 * s_ + hi) ^ sw) << 4
                a_lane = (wm * mi_n * 32 + c) * 128
                b_lane = a_bytes + (wn * ni * 32 + c) * 128
                for mi in range(mi_n):
                    off = a_lane + g_off + mi * 4096
                    cell = lds[off // 16]
                    assert cell[0] == "a" and cell[1] == (wm * mi_n + mi) * 32 + c and cell[2] == 2 * s_ + hi
                    fa[mi][lane] = cell[3]
                    addr_a[mi][lane] = off
                for n_ in range(ni):
                    off = b_lane + g_off + n_ * 4096
                    cell = lds[off // 16]
                    assert cell[0] == "w" and cell[1] == (wn * ni + n_) * 32 + c and cell[2] == 2 * s_ + hi
                    fb[n_][lane] = cell[3]
                    addr_b[n_][lane] = off
            for lst in addr_a + addr_b:
                conflicts += bank_conflicts(lst)
            for mi in range(mi_n):
                for n_ in range(ni):
                    for lb in range(64):                                     # B-operand lane = pixel column of D
                        cb, hb = lb & 31, lb >> 5
                        for r in range(16):
                            row = (r & 3) + 8 * (r >> 2) + 4 * hb
                            acc[mi, n_, lb, r] += sum(float(np.dot(fb[n_][row + 32 * hh], fa[mi][cb + 32 * hh])) for hh in range(2))
        for mi in range(mi_n):
            for n_ in range(ni):
                scr = np.full((32, 36), np.nan)
                for lane in range(64):
                    c, hi = lane & 31, lane >> 5
                    for q in range(4):
                        scr[c, 8 * q + 4 * hi: 8 * q + 4 * hi + 4] = acc[mi, n_, lane, 4 * q: 4 * q + 4]
                mrow0, nf0 = (wm * mi_n + mi) * 32, (wn * ni + n_) * 32
                if not out_f32:
                    for it in range(2):
                        for lane in range(64):
                            row, c8 = it * 16 + (lane >> 2), lane & 3
                            out[mrow0 + row, nf0 + c8 * 8: nf0 + c8 * 8 + 8] = scr[row, c8 * 8: c8 * 8 + 8]
                else:
                    for it in range(4):
                        for lane in range(64):
                            row, c4 = it * 8 + (lane >> 3), lane & 7
                            out[mrow0 + row, nf0 + c4 * 4: nf0 + c4 * 4 + 4] = scr[row, c4 * 4: c4 * 4 + 4]
    return A, W, out, conflicts


@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("mi,ni,wm,wn", [(2, 5, 4, 2), (2, 4, 4, 2), (2, 2, 4, 2), (1, 5, 4, 2)])
def test_gemm_bf16y_tile_reproduces_a_times_w_transposed(mi, ni, wm, wn, out_f32):
    rng = np.random.default_rng(10 * mi + ni)
    A, W, out, conflicts = emulate_bf16y(mi, ni, wm, wn, rng, out_f32)
    ref = A @ W.T
    assert np.allclose(out, ref, rtol=1e-12, atol=1e-12), f"max |diff| = {np.abs(out - ref).max()}"
    assert conflicts == 0, f"{conflicts} extra LDS cycles from bank conflicts in the fragment reads"
