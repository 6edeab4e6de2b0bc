// k_gemm_epi.hpp -- the fp32 output epilogue shared by the 8-wave large-tile kernels whose accumulators have the
// 16x16 MFMA D layout (lane (c = lane & 15, g = lane >> 4) holds D[4g .. 4g+3][c], D rows = output channels, D columns =
// pixels): k_gemm2x.hip (v_mfma_f32_16x16x4_f32) and k_gemm3x.hip (v_mfma_f32_16x16x32_bf16 on split operands).
//   * bias + time-embedding row + residual, or the GEGLU product, or a raw split-K slab
//   * every wave transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free by then)
//     and writes whole row segments with 16-byte lanes; the residual is read the same way
//   * split-K: raw fp32 slabs; launch_splitk_reduce -- or the GroupNorm / LayerNorm that reads the result (k_norm.hip) -- combines them
#pragma once
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float epi_f32x4 __attribute__((ext_vector_type(4)));

template <int MI, int NI, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_f32(const ConvGemm& p, epi_f32x4 (&acc)[MI][NI], unsigned char* smem_x32, const int m0,
                                                  const int n0, const int z, const int lid, const int wave, const int lane,
                                                  const int HoWo) {
    typedef epi_f32x4 f32x4;
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    const bool geglu = p.geglu != 0;
    // ---- epilogue: bias + time-embedding row + residual, fp32 -------------------------------------------
    // Each wave transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free
    // now) and writes whole 320-byte row segments with 16-byte lanes; the residual is read the same way.
    const bool split = p.splits > 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = !split && p.resid;
    const bool vec_ok = ((p.N & 3) == 0) && ((ldc & 3) == 0) && ((p.ldr & 3) == 0 || !has_resid);
    constexpr int LDSW = WNC + 4;       // scratch row stride in floats
    if (p.geglu == 2) {
        // GEGLU with the value and gate columns in DIFFERENT waves (round 5; tiles with an odd fragment count per wave, e.g. 256 x 160: 80 outputs per tile): wave
        // columns [0, WN / 2) computed x W_value, [WN / 2, WN) x W_gate for the same 16 NI outputs.  Per 16-row fragment group both waves of a pair put their
        // fragments (+ bias) into their scratch, the workgroup meets at a barrier, and each wave of the pair gates 8 of the 16 rows -- value from one scratch, gate
        // from the other -- and writes them as fp32 and / or planes.  Launch-side guarantees: no split-K, N % 4 == 0, no rowvec / residual.
        if constexpr (WN % 2 == 0) {
            constexpr int HWN = WN / 2;
            const int vg = wn / HWN, col = wn - vg * HWN;
            const int pw = wm * WN + (vg ? wn - HWN : wn + HWN);     // the partner wave
            __syncthreads();                                        // every wave is done with the last k tile
            float* scr = reinterpret_cast<float*>(smem_x32 + wave * (16 * LDSW * 4));
            const float* scr_v = reinterpret_cast<const float*>(smem_x32 + (vg ? pw : wave) * (16 * LDSW * 4));
            const float* scr_g = reinterpret_cast<const float*>(smem_x32 + (vg ? wave : pw) * (16 * LDSW * 4));
            const int nw0 = n0 + col * WNC;
            const float* bias = p.bias ? p.bias + (vg ? p.N : 0) : nullptr;
            f32x4 bias_g[NI];     // (round 6: the wave's bias columns once, in front of the fragment groups -- they were re-read behind each group's stores)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = nw0 + ni * 16 + g4 * 4;
                bias_g[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (bias && n < p.N) bias_g[ni] = *reinterpret_cast<const f32x4*>(bias + n);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = nw0 + ni * 16 + g4 * 4;
                    f32x4 v = acc[mi][ni];
                    if (bias && n < p.N) v += bias_g[ni];
                    *reinterpret_cast<f32x4*>(scr + c15 * LDSW + ni * 16 + g4 * 4) = v;
                }
                __syncthreads();
                constexpr int CH = WNC / 4;   // 16-byte chunks per row
#pragma unroll
                for (int q0 = 0; q0 < 8 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = 8 * vg + q / CH, c4 = q - (q / CH) * CH;
                    const int m = mrow0 + row, n = nw0 + c4 * 4;
                    if (q < 8 * CH && m < p.M && n < p.N) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(scr_v + row * LDSW + c4 * 4);
                        const f32x4 g = *reinterpret_cast<const f32x4*>(scr_g + row * LDSW + c4 * 4);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = v[e] * (0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752440f)));
                        if (p.C) *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = o;
                        if (p.C3) s3_store4(reinterpret_cast<unsigned char*>(p.C3) + (long long)m * p.ldc3, n, o);
                    }
                }
                __syncthreads();
            }
        }
        return;
    }
    if (geglu) {   // launch-side guarantees: NI even, no split-K, N % 8 == 0, ldc % 8 == 0, no rowvec / residual
        if constexpr (NI % 2 == 0) {
            constexpr int WNO = WNC / 2;     // output columns of a wave tile
            constexpr int LDSW2 = WNO + 4;
            __syncthreads();
            float* scr = reinterpret_cast<float*>(smem_x32 + wave * (16 * LDSW2 * 4));
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
                constexpr int CH = WNO / 4;
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c4 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c4 * 4;
                    if (q < 16 * CH && m < p.M && n < p.N)
                        *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c4 * 4);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    // ---- the lean form (round 6; ConvGemm::variant bit 6 switches it off), k_gemm_bf16_epi.hpp's for fp32 output: an INTERIOR tile whose rows belong to one sample (the
    // time-embedding row is then a per-column term like the bias) needs no bounds check, no sample-index division per fragment group and no 64-bit address product per
    // store.  Every lane-derived address is computed once per tile -- one LDS write address, NI LDS read addresses, NI 32-bit offsets each for the output, the residual and
    // the planes -- against wave-uniform row pointers that advance by 16 rows per group; the residual of group mi + 1 is requested before group mi is stored.  The general
    // form below executed ~3x the instructions per group and waited out every time-embedding load where it was issued.  Same values in the same order.
    if (!(p.variant & 64) && vec_ok && m0 + BM <= p.M && n0 + BN <= p.N) {
        int smp0 = 0;
        bool one_smp = true;
        if (!split && p.rowvec) {
            smp0 = m0 / HoWo;
            one_smp = (m0 + BM - 1) / HoWo == smp0;
        }
        if (one_smp) {
            typedef unsigned int u2 __attribute__((ext_vector_type(2)));
            const int wu = __builtin_amdgcn_readfirstlane(wave);
            const int wmu = wu / WN, wnu = wu - wmu * WN;
            constexpr int CH = WNC / 4;   // 16-byte chunks per row; 16 rows x CH chunks = NI x 64 lanes exactly
            __syncthreads();                // every wave is done with the last k tile
            float* scr = reinterpret_cast<float*>(smem_x32 + wu * (16 * LDSW * 4));
            const int nw0 = n0 + wnu * WNC;
            const bool has_c3 = !split && p.C3;
            f32x4 bias_v[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = nw0 + ni * 16 + g4 * 4;
                bias_v[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!split && p.bias) bias_v[ni] = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (!split && p.rowvec) bias_v[ni] += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp0 * p.rowvec_stride + n);
            }
            float* wr = scr + c15 * LDSW + g4 * 4;
            const float* rd[NI];
            unsigned goff[NI], roff[NI], poff[NI];
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                const int q = r * 64 + lane, row = q / CH, c4 = q - row * CH;
                rd[r] = scr + row * LDSW + c4 * 4;
                goff[r] = (unsigned)(row * ldc + c4 * 4) * 4u;
                roff[r] = has_resid ? (unsigned)(row * p.ldr + c4 * 4) * 4u : 0u;
                poff[r] = has_c3 ? (unsigned)(row * p.ldc3) + (unsigned)s3_plane_byte(nw0 + c4 * 4, 0) : 0u;
            }
            const long long mw0 = m0 + wmu * MI * 16;
            unsigned char* gbase = reinterpret_cast<unsigned char*>(Cf) + (mw0 * ldc + nw0) * 4;
            const unsigned char* rbase = reinterpret_cast<const unsigned char*>(p.resid) + (has_resid ? (mw0 * p.ldr + nw0) * 4 : 0);
            unsigned char* pbase = reinterpret_cast<unsigned char*>(p.C3) + (has_c3 ? mw0 * p.ldc3 : 0);
            f32x4 res[NI];
            if (has_resid) {
#pragma unroll
                for (int r = 0; r < NI; ++r) res[r] = *reinterpret_cast<const f32x4*>(rbase + roff[r]);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x4 v = acc[mi][ni];
                    if (!split) v += bias_v[ni];
                    *reinterpret_cast<f32x4*>(wr + ni * 16) = v;
                }
                __builtin_amdgcn_wave_barrier();
                f32x4 o[NI];
#pragma unroll
                for (int r = 0; r < NI; ++r) o[r] = *reinterpret_cast<const f32x4*>(rd[r]);
                __builtin_amdgcn_wave_barrier();
                if (has_resid) {
#pragma unroll
                    for (int r = 0; r < NI; ++r) o[r] += res[r];
                    if (mi + 1 < MI) {
                        rbase += (long long)p.ldr * 64;
#pragma unroll
                        for (int r = 0; r < NI; ++r) res[r] = *reinterpret_cast<const f32x4*>(rbase + roff[r]);
                    }
                }
                if (Cf) {
#pragma unroll
                    for (int r = 0; r < NI; ++r) *reinterpret_cast<f32x4*>(gbase + goff[r]) = o[r];
                    gbase += (long long)ldc * 64;
                }
                if (has_c3) {
#pragma unroll
                    for (int r = 0; r < NI; ++r) {
                        unsigned h[2], m[2], l[2];
                        s3_split4(o[r], h, m, l);
                        unsigned char* d = pbase + poff[r];
                        *reinterpret_cast<u2*>(d) = u2{h[0], h[1]};
                        *reinterpret_cast<u2*>(d + 64) = u2{m[0], m[1]};
                        *reinterpret_cast<u2*>(d + 128) = u2{l[0], l[1]};
                    }
                    pbase += (long long)p.ldc3 * 16;
                }
            }
            return;
        }
    }
    if (vec_ok) {
        __syncthreads();                // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_x32 + wave * (16 * LDSW * 4));
        const int nw0 = n0 + wn * WNC;
        constexpr int CH = WNC / 4;   // 16-byte chunks per row
        constexpr int NQ = (16 * CH + 63) / 64;     // row-coalesced 16-byte pieces per lane per fragment group
        // Round 6 (ConvGemm::variant bit 5 = off): every global LOAD of the epilogue is issued before its first fragment group is transposed -- the bias of the wave's
        // columns once (it was re-read for each of the MI groups: a store that may alias sits between two groups) and, on the tiles whose groups fit the registers
        // (MI NQ <= 8 pieces), the residual of ALL groups -- so the groups' load latencies overlap each other and the first LDS round trip instead of adding up.
        const bool early = !(p.variant & 32);
        constexpr bool PRE_R = MI * NQ <= 8;
        f32x4 bias_v[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = nw0 + ni * 16 + g4 * 4;
            bias_v[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (early && !split && p.bias && n < p.N) bias_v[ni] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
        f32x4 res_v[PRE_R ? MI : 1][PRE_R ? NQ : 1];
        if constexpr (PRE_R) {
            if (early && has_resid) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < NQ; ++r) {
                        const int q = r * 64 + lane;
                        const int row = q / CH, c4 = q - row * CH;
                        const int m = m0 + (wm * MI + mi) * 16 + row, n = nw0 + c4 * 4;
                        res_v[mi][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (q < 16 * CH && m < p.M && n < p.N) res_v[mi][r] = *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                    }
            }
        }
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
                        if (early) v += bias_v[ni];
                        else if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    }
                    *reinterpret_cast<f32x4*>(scr + c15 * LDSW + ni * 16 + g4 * 4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                const int q = r * 64 + lane;
                const int row = q / CH, c4 = q - row * CH;
                const int m = mrow0 + row, n = nw0 + c4 * 4;
                if (q < 16 * CH && m < p.M && n < p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                    if (has_resid) {
                        if (PRE_R && early) v += res_v[PRE_R ? mi : 0][PRE_R ? r : 0];
                        else v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                    }
                    if (split) *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;     // (Cf = this k slice's slab)
                    else {
                        if (Cf) *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
                        if (p.C3) s3_store4(reinterpret_cast<unsigned char*>(p.C3) + (long long)m * p.ldc3, n, v);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
    // odd strides / N not a multiple of 4: element-wise stores straight from the accumulators
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
                        if (p.resid) sv += p.resid[(long long)m * p.ldr + n + r];
                    }
                    Cf[(long long)m * ldc + n + r] = sv;
                }
            }
        }
    }
    }
}

}  // namespace sdmi
