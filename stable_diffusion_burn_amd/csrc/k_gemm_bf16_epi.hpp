// k_gemm_bf16_epi.hpp -- the epilogue shared by the large-tile bf16 GEMM kernels (k_gemm_bf16x.hip, k_gemm_bf16p.hip): fp32 bias +
// time-embedding row + (bf16) residual, then a bf16 or fp32 store (split-K: the k slice's fp32 slab), or the GEGLU gate.
// acc[mi][ni] is the 16x16 fragment of rows (wm MI + mi) 16 .., columns (wn NI + ni) 16 ..; lane (c = lane & 15, g = lane >> 4) holds
// columns 4 g .. 4 g + 3 of row c.  Every wave of the workgroup calls it exactly once (one __syncthreads() inside); smem_x is the
// kernel's LDS, free for reuse once every wave has passed that barrier and no LDS-DMA is in flight.
#pragma once
#include "kernels.hpp"

namespace sdmi {

typedef float bepi_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bepi_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bepi_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xbf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float xbf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned xf32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
typedef float bepi_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bepi_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of bf16 (a low, b high), round to nearest even: ONE v_cvt_pk_bf16_f32 (the integer form above is 4 instructions per value; the epilogue of a
// 256 x 320 tile packs 81 920 values -- profiles/r04ad_*: 8 of a short-K launch's 27 us per tile were this epilogue's instructions, not its stores)
__device__ __forceinline__ unsigned xpack_bf16x2(float a, float b) {
    const bepi_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bepi_bf16x2));
}

// v_permlane16_swap_b32: the odd 16-lane rows of a swap with the even rows of b (a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]).  Inline asm with its
// own wait states (a VALU write of either operand needs two before the swap reads it; hipcc pads nothing inside an asm string).
__device__ __forceinline__ void xswap16(unsigned& a, unsigned& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// two packed fragments a, b (two dwords each) -> the 16 bytes this lane stores: {a.x, a.y, b.x, b.y} after both dwords were exchanged
__device__ __forceinline__ bepi_u32x4 xswap16_pair(bepi_u32x2 a, bepi_u32x2 b) {
    unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    xswap16(a0, b0);
    xswap16(a1, b1);
    return bepi_u32x4{a0, a1, b0, b1};
}

// pre_synced: the caller has already passed a workgroup barrier behind the last k tile (the persistent kernel, which issues the next tile's first DMA between that
//             barrier and this epilogue) -- smem_x is then the stage the next tile does NOT land in.
// direct:     bf16 output without a residual leaves WITHOUT the LDS transpose (round 5): the packed fragments of two neighbouring column groups are exchanged between the
//             wave's 16-lane rows (two v_permlane16_swap per pair), after which every lane holds 8 consecutive channels of one row = one 16-byte store; a row receives
//             64-byte pieces instead of whole 160-byte segments, the values and their single rounding are the same.
// MODE: -1 = every decision below is taken at run time from p (the one-tile-per-workgroup kernels); 0 / 1 / 2 = bf16 output through the 16-byte paths, known at compile
//             time to be plain / with a residual / the GEGLU gate (the persistent kernel: inside its tile loop the run-time form with all its paths costs 200 spilled
//             registers) -- the launcher checks what the run-time form checks.
template <int MI, int NI, int WM, int WN, int MODE = -1>
__device__ __forceinline__ void gemm_epilogue_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], unsigned char* smem_x, const int m0, const int n0,
                                                   const int z, const int wave, const int lane, const int HoWo, const bool pre_synced = false,
                                                   const bool direct = false) {
    typedef bepi_f32x4 f32x4;
    typedef bepi_u32x4 u32x4;
    typedef bepi_u32x2 u32x2;
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const bool geglu = MODE < 0 ? p.geglu != 0 : MODE == 2;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual, then bf16 or fp32 store ----------
    // A lane holds 4 consecutive channels of 16 different rows, so storing straight from the accumulators issues
    // 8-byte pieces at a row stride (measured: 38k cycles for a 256x320 tile, store-issue bound).  Each wave instead
    // transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free now) and
    // writes whole 160-byte row segments with 16-byte lanes; the residual is read the same way.
    const bool split = MODE < 0 ? p.splits > 1 : false;
    const bool out_f32 = MODE < 0 ? (split || p.out_mode == 1) : false;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = MODE < 0 ? (!split && p.resid) : MODE == 1;
    const bool vec_ok = MODE < 0 ? (((p.N & 7) == 0) && ((ldc & 7) == 0) && ((p.ldr & 7) == 0 || !has_resid)) : true;
    constexpr int LDSW = WNC + 4;       // scratch row stride in floats (336 B for NI = 5: 16-byte aligned, rows on distinct banks)
    if (geglu) {   // launch-side guarantees: NI even, no split-K, N % 8 == 0, ldc % 8 == 0, no rowvec / residual
        if constexpr (NI % 2 == 0) {
            constexpr int WNO = WNC / 2;     // output columns of a wave tile
            constexpr int LDSW2 = WNO + 4;
            if (!pre_synced) __syncthreads();
            float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW2 * 4));
            const int nw0 = n0 + wn * WNO;
            auto gate = [&](int mi, int j) {
                const int n = nw0 + j * 16 + g4 * 4;
                f32x4 v = acc[mi][2 * j], g = acc[mi][2 * j + 1];
                if (p.bias && n < p.N) {
                    v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    g += *reinterpret_cast<const f32x4*>(p.bias + p.N + n);
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[e] * (0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752440f)));
                return o;
            };
            constexpr int CH = WNO / 8;
            if (direct && (MODE >= 0 || p.out_mode != 1)) {
                constexpr int NJ = NI / 2;
                static_assert(NJ % 2 == 0 || MI % 2 == 0, "an odd fragment column pairs with the next row group");
                const int gl = g4 & 1, gh = g4 >> 1;
                unsigned short* Cg = reinterpret_cast<unsigned short*>(p.C);
                auto packed = [&](int mi, int j) {
                    const f32x4 o = gate(mi, j);
                    return u32x2{xpack_bf16x2(o[0], o[1]), xpack_bf16x2(o[2], o[3])};
                };
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = m0 + (wm * MI + mi) * 16 + c15;
#pragma unroll
                    for (int j = 0; j + 1 < NJ; j += 2) {
                        const u32x2 a = packed(mi, j), b = packed(mi, j + 1);
                        const u32x4 o4 = xswap16_pair(a, b);
                        const int n = nw0 + (j + gl) * 16 + gh * 8;
                        if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Cg + (long long)m * p.ldc + n) = o4;
                    }
                    if ((NJ & 1) && !(mi & 1)) {
                        const u32x2 a = packed(mi, NJ - 1), b = packed(mi + 1, NJ - 1);
                        const u32x4 o4 = xswap16_pair(a, b);
                        const int mm = m + gl * 16, n = nw0 + (NJ - 1) * 16 + gh * 8;
                        if (mm < p.M && n < p.N) *reinterpret_cast<u32x4*>(Cg + (long long)mm * p.ldc + n) = o4;
                    }
                }
            } else if (MODE >= 0 || p.out_mode != 1) {
                // bf16 output: rounded before the transpose, 2-byte scratch (see the main path below), the gate of group mi + 1 behind the stores of group mi
                constexpr int RSB = WNO * 2 + 16;
                static_assert(((RSB / 16) & 1) == 1, "scratch rows must start on distinct 16-byte bank slots");
                constexpr int NR = (16 * CH + 63) / 64;
                unsigned char* sb = smem_x + wave * (16 * LDSW2 * 4);
                auto stage_b = [&](int mi) {
#pragma unroll
                    for (int j = 0; j < NI / 2; ++j) {
                        const f32x4 o = gate(mi, j);
                        const u32x2 w = {xpack_bf16x2(o[0], o[1]), xpack_bf16x2(o[2], o[3])};
                        *reinterpret_cast<u32x2*>(sb + c15 * RSB + j * 32 + g4 * 8) = w;
                    }
                };
                stage_b(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int mrow0 = m0 + (wm * MI + mi) * 16;
                    u32x4 o[NR];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int q = r * 64 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        if (q < 16 * CH) o[r] = *reinterpret_cast<const u32x4*>(sb + row * RSB + c8 * 16);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (mi + 1 < MI) stage_b(mi + 1);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int q = r * 64 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        const int m = mrow0 + row, n = nw0 + c8 * 8;
                        if (q < 16 * CH && m < p.M && n < p.N) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o[r];
                    }
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int mrow0 = m0 + (wm * MI + mi) * 16;
#pragma unroll
                    for (int j = 0; j < NI / 2; ++j) *reinterpret_cast<f32x4*>(scr + c15 * LDSW2 + j * 16 + g4 * 4) = gate(mi, j);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                        const int q = q0 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        const int m = mrow0 + row, n = nw0 + c8 * 8;
                        if (q < 16 * CH && m < p.M && n < p.N) {
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8 + 4);
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = lo;
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n + 4) = hi;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        return;
    }
    if (vec_ok && direct && !out_f32 && !has_resid) {
        static_assert(NI % 2 == 0 || MI % 2 == 0, "an odd fragment column pairs with the next row group");
        const int nw0 = n0 + wn * WNC;
        const int gl = g4 & 1, gh = g4 >> 1;
        auto packed = [&](int mi, int ni) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
            const int smp = (m < p.M ? m : 0) / HoWo;
            const int n = nw0 + ni * 16 + g4 * 4;
            f32x4 v = acc[mi][ni];
            if (n < p.N) {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
            }
            return u32x2{xpack_bf16x2(v[0], v[1]), xpack_bf16x2(v[2], v[3])};
        };
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
#pragma unroll
            for (int j = 0; j + 1 < NI; j += 2) {
                // lane rows after the exchange: r0 = (ni j, columns 0..7), r1 = (ni j + 1, 0..7), r2 = (ni j, 8..15), r3 = (ni j + 1, 8..15) of tile row c15
                const u32x2 a = packed(mi, j), b = packed(mi, j + 1);
                const u32x4 o4 = xswap16_pair(a, b);
                const int n = nw0 + (j + gl) * 16 + gh * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o4;
            }
            if ((NI & 1) && !(mi & 1)) {   // the odd column group of row groups mi and mi + 1 share their stores
                const u32x2 a = packed(mi, NI - 1), b = packed(mi + 1, NI - 1);
                const u32x4 o4 = xswap16_pair(a, b);
                const int mm = m + gl * 16, n = nw0 + (NI - 1) * 16 + gh * 8;
                if (mm < p.M && n < p.N) *reinterpret_cast<u32x4*>(Ch + (long long)mm * ldc + n) = o4;
            }
        }
    } else if (vec_ok && direct && !out_f32) {
        // ... with a residual: the add precedes the only rounding, so the fp32 values are exchanged (four swaps per fragment pair), after which a lane holds 8 consecutive
        // channels of one row, reads the residual's 16 bytes there, adds, rounds and stores
        static_assert(NI % 2 == 0 || MI % 2 == 0, "an odd fragment column pairs with the next row group");
        const int nw0 = n0 + wn * WNC;
        const int gl = g4 & 1, gh = g4 >> 1;
        auto biased = [&](int mi, int ni) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
            const int smp = (m < p.M ? m : 0) / HoWo;
            const int n = nw0 + ni * 16 + g4 * 4;
            f32x4 v = acc[mi][ni];
            if (n < p.N) {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
            }
            return v;
        };
        auto emit = [&](f32x4 a, f32x4 b, int m, int n) {
            unsigned a0 = __float_as_uint(a[0]), a1 = __float_as_uint(a[1]), a2 = __float_as_uint(a[2]), a3 = __float_as_uint(a[3]);
            unsigned b0 = __float_as_uint(b[0]), b1 = __float_as_uint(b[1]), b2 = __float_as_uint(b[2]), b3 = __float_as_uint(b[3]);
            xswap16(a0, b0); xswap16(a1, b1); xswap16(a2, b2); xswap16(a3, b3);
            if (m < p.M && n < p.N) {
                const u32x4 rr = *reinterpret_cast<const u32x4*>(Rh + (long long)m * p.ldr + n);
                const u32x4 o = {xpack_bf16x2(__uint_as_float(a0) + xbf16_lo(rr[0]), __uint_as_float(a1) + xbf16_hi(rr[0])),
                                 xpack_bf16x2(__uint_as_float(a2) + xbf16_lo(rr[1]), __uint_as_float(a3) + xbf16_hi(rr[1])),
                                 xpack_bf16x2(__uint_as_float(b0) + xbf16_lo(rr[2]), __uint_as_float(b1) + xbf16_hi(rr[2])),
                                 xpack_bf16x2(__uint_as_float(b2) + xbf16_lo(rr[3]), __uint_as_float(b3) + xbf16_hi(rr[3]))};
                *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o;
            }
        };
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
#pragma unroll
            for (int j = 0; j + 1 < NI; j += 2) emit(biased(mi, j), biased(mi, j + 1), m, nw0 + (j + gl) * 16 + gh * 8);
            if ((NI & 1) && !(mi & 1)) emit(biased(mi, NI - 1), biased(mi + 1, NI - 1), m + gl * 16, nw0 + (NI - 1) * 16 + gh * 8);
        }
    } else if (vec_ok) {
        if (!pre_synced) __syncthreads();                // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW * 4));
        const int nw0 = n0 + wn * WNC;
        // fragment group mi (16 rows x WNC columns of this wave) -> scratch, with the per-column / per-sample terms
        auto stage = [&](int mi) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
            const int smp = (m < p.M ? m : 0) / HoWo;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = nw0 + ni * 16 + g4 * 4;
                f32x4 v = acc[mi][ni];
                if (!split && n < p.N) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                }
                *reinterpret_cast<f32x4*>(scr + c15 * LDSW + ni * 16 + g4 * 4) = v;
            }
        };
        if (!out_f32 && !has_resid) {
            // bf16 output without a residual: the values are rounded to bf16 BEFORE the transpose (the same single rounding), so the scratch holds 2 bytes per
            // element -- 5 ds_write_b64 + 3 ds_read_b128 per fragment group where the fp32 scratch below takes 5 ds_write_b128 + 6 ds_read_b128 whose 32-byte lane
            // stride uses half the banks.  Row stride WNC 2 + 16 bytes = 4 x odd dwords: the 16 rows of a half-wave's ds_write_b64 tile the 64 banks.
            constexpr int RSB = WNC * 2 + 16;
            static_assert(((RSB / 16) & 1) == 1, "scratch rows must start on distinct 16-byte bank slots");
            constexpr int CH = WNC / 8;   // 16-byte bf16 chunks per row
            constexpr int NR = (16 * CH + 63) / 64;
            unsigned char* sb = smem_x + wave * (16 * LDSW * 4);
            auto stage_b = [&](int mi) {
                const int m = m0 + (wm * MI + mi) * 16 + c15;
                const int smp = (m < p.M ? m : 0) / HoWo;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = nw0 + ni * 16 + g4 * 4;
                    f32x4 v = acc[mi][ni];
                    if (n < p.N) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    }
                    const u32x2 w = {xpack_bf16x2(v[0], v[1]), xpack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(sb + c15 * RSB + ni * 32 + g4 * 8) = w;
                }
            };
            stage_b(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                u32x4 o[NR];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    if (q < 16 * CH) o[r] = *reinterpret_cast<const u32x4*>(sb + row * RSB + c8 * 16);
                }
                __builtin_amdgcn_wave_barrier();
                if (mi + 1 < MI) stage_b(mi + 1);    // (in-order LDS: lands behind the reads above)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o[r];
                }
            }
        } else if (!out_f32) {
            // bf16 output.  Software-pipelined over the fragment groups: group mi is read back row-coalesced into registers, group mi + 1 is staged into the SAME
            // scratch right behind those reads (the LDS executes one wave's instructions in order, so the writes land after them), and only then are the rows of
            // group mi converted and stored -- the LDS round trip of the next group hides behind the conversion and the store issue of this one
            constexpr int CH = WNC / 8;   // 16-byte bf16 chunks per row
            constexpr int NR = (16 * CH + 63) / 64;
            stage(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                f32x4 lo[NR], hi[NR];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    if (q < 16 * CH) {
                        lo[r] = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8);
                        hi[r] = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8 + 4);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (mi + 1 < MI) stage(mi + 1);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        f32x4 l = lo[r], h = hi[r];
                        if (has_resid) {
                            const u32x4 rr = *reinterpret_cast<const u32x4*>(Rh + (long long)m * p.ldr + n);
                            l[0] += xbf16_lo(rr[0]); l[1] += xbf16_hi(rr[0]); l[2] += xbf16_lo(rr[1]); l[3] += xbf16_hi(rr[1]);
                            h[0] += xbf16_lo(rr[2]); h[1] += xbf16_hi(rr[2]); h[2] += xbf16_lo(rr[3]); h[3] += xbf16_hi(rr[3]);
                        }
                        const u32x4 o = {xpack_bf16x2(l[0], l[1]), xpack_bf16x2(l[2], l[3]), xpack_bf16x2(h[0], h[1]), xpack_bf16x2(h[2], h[3])};
                        *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                stage(mi);
                __builtin_amdgcn_wave_barrier();
                constexpr int CH = WNC / 4;   // 16-byte fp32 chunks per row
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c4 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c4 * 4;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                        if (has_resid) {
                            const u32x2 r = *reinterpret_cast<const u32x2*>(Rh + (long long)m * p.ldr + n);
                            v[0] += xbf16_lo(r[0]); v[1] += xbf16_hi(r[0]); v[2] += xbf16_lo(r[1]); v[3] += xbf16_hi(r[1]);
                        }
                        *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;     // (split-K: Cf = this k slice's fp32 slab)
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
    // odd strides / N not a multiple of 8: element-wise stores straight from the accumulators
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        if (m >= p.M) continue;
        const int smp = m / HoWo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 16 + g4 * 4;
            if (n >= p.N) continue;
            const f32x4 v = acc[mi][ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < p.N) {
                    float sv = v[r];
                    if (!split) {
                        if (p.bias) sv += p.bias[n + r];
                        if (p.rowvec) sv += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                        if (p.resid) sv += __uint_as_float((unsigned)Rh[(long long)m * p.ldr + n + r] << 16);
                    }
                    if (out_f32) Cf[(long long)m * ldc + n + r] = sv;
                    else Ch[(long long)m * ldc + n + r] = (unsigned short)xf32_to_bf16_bits(sv);
                }
            }
        }
    }
    }
}

}  // namespace sdmi
