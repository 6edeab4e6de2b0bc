// k_gemm_bf16x.hip -- large-tile bf16 implicit-GEMM conv / linear for precision = 1 at batch sizes where the
// GEMMs are big (BASELINE.json configs[2..3]: M = n*Ho*Wo in the tens of thousands).
//
// Why a second bf16 kernel: at the bf16 matrix rate (16 cycles per v_mfma_f32_16x16x32_bf16) the 4-wave
// 128x128 structure inherited from the fp32 kernel is LDS-bound -- per 64-deep k tile it writes 32 KB through
// the VGPR->LDS path (~79 B/clk) and reads 16 fragments per wave, ~670 LDS cycles against 512 MFMA cycles.
// This kernel follows the MI355X GEMM recipe instead:
//   * 256-row tiles, 8 waves (512 threads), one workgroup per CU: wave tiles of 128x80 / 128x64 / 64x64
//     halve the LDS bytes per flop;
//   * operands are staged HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no
//     staging registers, no ds_write pass.  The DMA writes lane-linearly (wave base + lane*16), so the
//     XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied on the SOURCE side:
//     LDS slot (row r, chunk s) receives global chunk s ^ (r & 7) of that row -- still the same full 128-byte
//     line per row;
//   * zero fill (conv padding taps, M / N tails) by pointing the lane at a zero page instead of predication,
//     so every lane always issues its DMA;
//   * two LDS stages, one __syncthreads() per k tile: [barrier: tile t landed, stage t^1 free] -> issue the
//     DMA of tile t+1 -> 2 x (fragment reads + MFMAs) on tile t.  The DMA is in flight during the whole MFMA
//     phase; the barrier's vmcnt(0) retires it.
// k order, weight packing, swapped MFMA operands (a lane holds 4 consecutive output channels), XCD-aware tile
// map, deterministic split-K and the fused epilogue are those of k_gemm_bf16.hip.
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

__device__ __forceinline__ float xbf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float xbf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned xf32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned xpack_bf16x2(float a, float b) { return xf32_to_bf16_bits(a) | (xf32_to_bf16_bits(b) << 16); }

static const GemmTileInfo kTilesX[kNumGemmTilesX] = {
    {256, 320, "256x320x"}, {256, 256, "256x256x"}, {256, 128, "256x128x"}, {128, 320, "128x320x"}};
const GemmTileInfo& gemm_tile_info_x(int cfg) { return kTilesX[cfg]; }

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm_bf16x_kernel(const ConvGemm p) {
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "every wave issues whole 8-row DMA pieces");
    constexpr int NA = BM / 64;               // A pieces (8 rows x 128 B) per wave per k tile
    constexpr int NB = BN / 64;               // B pieces per wave
    constexpr int STAGE = (BM + BN) * 128;    // bytes of one LDS stage

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    // GEGLU mode: a tile pairs BN/2 value columns with their BN/2 gate columns (fragment ni even = values, odd = gates of
    // the same outputs), so a lane holds both and the epilogue emits value * gelu(gate)
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const bool geglu = p.geglu != 0;
    const int BNO = geglu ? BN / 2 : BN;   // output columns per tile
    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BNO - 1) / BNO;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int lid = gw.lid;
    const int tm = gw.tm;
    const int tn = gw.tn;
    const int m0 = tm * BM;
    const int n0 = tn * BNO;

    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    f32x4 acc[MI][NI];
    const int c15 = lane & 15, g4 = lane >> 4;
    {
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const long long pix_bytes = (long long)p.a_ld * 2;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // DMA piece j of a wave covers tile rows (wave + 8 j) * 8 .. + 7; lane -> row + (lane >> 3), LDS slot lane & 7,
    // which receives global chunk (lane & 7) ^ (row & 7) = (lane & 7) ^ (lane >> 3)
    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ sub;

    int a_iy0[NA], a_ix0[NA];
    long long a_nboff[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave + 8 * j) * 8 + sub;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_nboff[j] = (long long)nb * (p.Hs * p.Ws) * pix_bytes + chunk * 16;
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
        a_ix0[j] = ox * p.stride - p.pad;
    }
    const char* b_src[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int r0 = (wave + 8 * j) * 8 + sub;   // tile row of operand B
        int n = n0 + r0;
        long long wrow = n;
        if (geglu) {
            const int f = r0 >> 4, fw = f / NI, ni = f - fw * NI;
            n = n0 + fw * (WNC / 2) + (ni >> 1) * 16 + (r0 & 15);
            wrow = (long long)n + ((ni & 1) ? p.N : 0);
        }
        b_src[j] = (n < p.N) ? Bbase + wrow * p.b_ld * 2 + chunk * 16 : nullptr;
    }

    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;

    auto issue = [&](int buf) {   // DMA of k tile kt_next into LDS stage buf; advances (cs, ky, kx)
        unsigned char* stage = smem_x + buf * STAGE;
        const long long c0b = (long long)cs * 128;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int iy = a_iy0[j] + ky;
            const int ix = a_ix0[j] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const long long pix = (long long)((iy >> p.ups) * p.Ws + (ix >> p.ups));
            const char* src = ok ? Abase + a_nboff[j] + pix * pix_bytes + c0b : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        const long long k0b = (long long)kt_next * 128;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const char* src = b_src[j] ? b_src[j] + k0b : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + BM * 128 + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        const bool wrap_x = (kx + 1 == p.KW);
        const bool wrap_y = wrap_x && (ky + 1 == p.KH);
        kx = wrap_x ? 0 : kx + 1;
        ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
        cs = wrap_y ? cs + 1 : cs;
        ++kt_next;
    };

    // fragment reads: lane (c = lane & 15, g = lane >> 4) reads row base + c, chunk (4 kk + g) ^ (c & 7)
    const int fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    const int fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
    const int a_base = wm * 16 * MI * 128;
    const int b_base = BM * 128 + wn * 16 * NI * 128;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        if (t + 1 < n_t) issue(cur ^ 1);
        const unsigned char* stage = smem_x + cur * STAGE;
        // Fragment reads run one group of rows ahead of the MFMAs that use them: the reads of rows [g+1] are issued
        // before the MFMAs of rows [g], so the compiler's counted lgkmcnt waits find the data already there
        // (one ds_read_b128 per 5 MFMAs; issuing them two at a time right before use stalls every 10 MFMAs).
        constexpr int GM = (MI >= 8) ? 4 : 2;    // rows per group
        constexpr int NG = MI / GM;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? fr_off1 : fr_off0;
            u32x4 fb[NI];
            u32x4 fa[2][GM];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(stage + b_base + ni * 2048 + fo);
#pragma unroll
            for (int i = 0; i < GM; ++i) fa[0][i] = *reinterpret_cast<const u32x4*>(stage + a_base + i * 2048 + fo);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < GM; ++i)
                        fa[(g + 1) & 1][i] = *reinterpret_cast<const u32x4*>(stage + a_base + ((g + 1) * GM + i) * 2048 + fo);
                }
#pragma unroll
                for (int i = 0; i < GM; ++i)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[g * GM + i][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]),
                                                                                  __builtin_bit_cast(bf16x8, fa[g & 1][i]),
                                                                                  acc[g * GM + i][ni], 0, 0, 0);
            }
            // pin the issue order the source spells out (hipcc otherwise sinks each read next to its first use to save
            // registers): operand B and the first row group, then one read of the next group per row of MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, NI + GM, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < GM; ++i) {
                    if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                }
        }
    }
    }   // plain k loop

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

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg_bf16x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16x_kernel<MI, NI, WM, WN>;
    constexpr size_t lds = 2 * (size_t)(16 * MI * WM + 16 * NI * WN) * 128;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_bf16x(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesX) return hipErrorInvalidValue;
    if ((p.Cin % 64) || !p.zero_page) return hipErrorInvalidValue;
    if (p.geglu && (cfg == 0 || cfg == 3 || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;  // needs an even NI
    const int bm = kTilesX[cfg].bm, bn = kTilesX[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const int tiles = MT * NT;
    const dim3 grid = gemm_grid(p, tiles);
    switch (cfg) {
        case 0: return launch_cfg_bf16x<8, 5, 2, 4>(p, grid, stream);
        case 1: return launch_cfg_bf16x<8, 4, 2, 4>(p, grid, stream);
        case 2: return launch_cfg_bf16x<4, 4, 4, 2>(p, grid, stream);
        case 3: return launch_cfg_bf16x<4, 5, 2, 4>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
