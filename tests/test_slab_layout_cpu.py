"""The split-K slab layout of the plane GEMM (ConvGemm::slab_native): what a wave stores (k_gemm_epi.hpp) and what the combine kernel decodes
(splitk_reduce_native_kernel, k_gemm.hip) must name the same output element, cover every (m < M, n < N) exactly once, and stay inside the padded slab.
Both index computations are restated here from the kernels' source; the GPU parity tests (bit-identical results with slab_native = 0 / 1) check the kernels."""
import numpy as np
import pytest

# (MI, NI, WM, WN) of the k_gemm3p.hip tiles 300 + cfg (kShapeP)
SHAPES = [(4, 5, 4, 2), (4, 4, 4, 2), (4, 4, 2, 4), (2, 5, 4, 2), (2, 4, 4, 2), (2, 2, 2, 2), (2, 4, 2, 2), (4, 5, 1, 4), (2, 4, 4, 1)]


def store_slots(M, N, MI, NI, WM, WN):
    """slot index and (m, n) of every float4 a workgroup's waves store: gemm_epilogue_f32, slab_native branch"""
    BM, BN = 16 * MI * WM, 16 * NI * WN
    MT, NT = -(-M // BM), -(-N // BN)
    out = {}
    for tm in range(MT):
        for tn in range(NT):
            m0, n0 = tm * BM, tn * BN
            tile = (m0 // BM) * NT + n0 // BN
            for wave in range(WM * WN):
                wm, wn = wave // WN, wave % WN
                for mi in range(MI):
                    for ni in range(NI):
                        lane = np.arange(64)
                        slot = (tile * (WM * WN) + wave) * (MI * NI * 64) + (mi * NI + ni) * 64 + lane
                        m = m0 + (wm * MI + mi) * 16 + (lane & 15)          # accumulator layout: lane (c = lane & 15, g = lane >> 4) holds columns 4 g .. 4 g + 3 of row c
                        n = n0 + (wn * NI + ni) * 16 + (lane >> 4) * 4
                        for s, a, b in zip(slot, m, n):
                            assert s not in out
                            out[int(s)] = (int(a), int(b))
    return out, MT * NT * BM * BN // 4


def decode_slot(i, M, N, MI, NI, WM, WN):
    """splitk_reduce_native_kernel"""
    BM, BN = 16 * MI * WM, 16 * NI * WN
    NT = -(-N // BN)
    frags, waves = MI * NI, WM * WN
    lane = i & 63
    r = i >> 6
    frag = r % frags
    r //= frags
    wave = r % waves
    tile = r // waves
    tm, tn = tile // NT, tile % NT
    mi, ni = frag // NI, frag % NI
    wm, wn = wave // WN, wave % WN
    return tm * BM + (wm * MI + mi) * 16 + (lane & 15), tn * BN + (wn * NI + ni) * 16 + (lane >> 4) * 4


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mn", [(128, 320), (200, 132), (512, 1280), (77, 64)])
def test_store_and_combine_agree(shape, mn):
    M, N = mn
    stored, slots = store_slots(M, N, *shape)
    assert len(stored) == slots and max(stored) == slots - 1          # dense and inside the padded slab (slab_stride = 4 slots floats)
    live = set()
    for i, (m, n) in stored.items():
        assert decode_slot(i, M, N, *shape) == (m, n)
        if m < M and n < N:
            for e in range(4):
                assert n + e < N or N % 4                               # (the 16-byte path needs N % 4 == 0)
                live.add((m, n + e))
    assert len(live) == M * (N // 4 * 4)
