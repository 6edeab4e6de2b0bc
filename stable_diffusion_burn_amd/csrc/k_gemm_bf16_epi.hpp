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
__device__ __forceinline__ unsigned xpack_bf16x2(float a, float b) { return xf32_to_bf16_bits(a) | (xf32_to_bf16_bits(b) << 16); }

template <int MI, int NI, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], unsigned char* smem_x, const int m0, const int n0,
                                                   const int z, const int wave, const int lane, const int HoWo) {
    typedef bepi_f32x4 f32x4;
    typedef bepi_u32x4 u32x4;
    typedef bepi_u32x2 u32x2;
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const bool geglu = p.geglu != 0;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual, then bf16 or fp32 store ----------
    // A lane holds 4 consecutive channels of 16 different rows, so storing straight from the accumulators issues
    // 8-byte pieces at a row stride (measured: 38k cycles for a 256x320 tile, store-issue bound).  Each wave instead
    // transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free now) and
    // writes whole 160-byte row segments with 16-byte lanes; the residual is read the same way.
    const bool split = p.splits > 1;
    const bool out_f32 = split || p.out_mode == 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = !split && p.resid;
    const bool vec_ok = ((p.N & 7) == 0) && ((ldc & 7) == 0) && ((p.ldr & 7) == 0 || !has_resid);
    constexpr int LDSW = WNC + 4;       // scratch row stride in floats (336 B for NI = 5: 16-byte aligned, rows on distinct banks)
    if (geglu) {   // launch-side guarantees: NI even, no split-K, N % 8 == 0, ldc % 8 == 0, no rowvec / residual
        if constexpr (NI % 2 == 0) {
            constexpr int WNO = WNC / 2;     // output columns of a wave tile
            constexpr int LDSW2 = WNO + 4;
            __syncthreads();
            float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW2 * 4));
            const int nw0 = n0 + wn * WNO;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
#pragma unroll
                for (int j = 0; j < NI / 2; ++j) {
                    const int n = nw0 + j * 16 + g4 * 4;
                    f32x4 v = acc[mi][2 * j], g = acc[mi][2 * j + 1];
                    if (p.bias && n < p.N) {
                        v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        g += *reinterpret_cast<const f32x4*>(p.bias + p.N + n);
                    }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[e] * (0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752440f)));
                    *reinterpret_cast<f32x4*>(scr + c15 * LDSW2 + j * 16 + g4 * 4) = o;
                }
                __builtin_amdgcn_wave_barrier();
                constexpr int CH = WNO / 8;
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8);
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8 + 4);
                        if (p.out_mode == 1) {
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = lo;
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n + 4) = hi;
                        } else {
                            const u32x4 o = {xpack_bf16x2(lo[0], lo[1]), xpack_bf16x2(lo[2], lo[3]), xpack_bf16x2(hi[0], hi[1]), xpack_bf16x2(hi[2], hi[3])};
                            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    if (vec_ok) {
        __syncthreads();                // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW * 4));
        const int nw0 = n0 + wn * WNC;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mrow0 = m0 + (wm * MI + mi) * 16;
            {
                const int m = mrow0 + c15;
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
            }
            __builtin_amdgcn_wave_barrier();
            if (!out_f32) {
                constexpr int CH = WNC / 8;   // 16-byte bf16 chunks per row
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        f32x4 lo = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8);
                        f32x4 hi = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8 + 4);
                        if (has_resid) {
                            const u32x4 r = *reinterpret_cast<const u32x4*>(Rh + (long long)m * p.ldr + n);
                            lo[0] += xbf16_lo(r[0]); lo[1] += xbf16_hi(r[0]); lo[2] += xbf16_lo(r[1]); lo[3] += xbf16_hi(r[1]);
                            hi[0] += xbf16_lo(r[2]); hi[1] += xbf16_hi(r[2]); hi[2] += xbf16_lo(r[3]); hi[3] += xbf16_hi(r[3]);
                        }
                        const u32x4 o = {xpack_bf16x2(lo[0], lo[1]), xpack_bf16x2(lo[2], lo[3]), xpack_bf16x2(hi[0], hi[1]), xpack_bf16x2(hi[2], hi[3])};
                        *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o;
                    }
                }
            } else {
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
            }
            __builtin_amdgcn_wave_barrier();
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
