// k_fp8.hip -- precision = 2 (BASELINE.json configs[4], "fp8 conv"): the ResBlock / ResnetBlock 3x3 convolutions
// (unet/mod.rs:713-733, autoencoder/mod.rs:514-527) on the block-scaled MX matrix instruction of gfx950,
//   v_mfma_scale_f32_16x16x128_f8f6f4  (e4m3 x e4m3, one E8M0 scale per 32 consecutive k, fp32 accumulate),
// which runs at twice the bf16 rate (~5 PFLOP/s dense).  Everything else of the model stays on the bf16 path.
//
// Operand layout of the instruction, measured with tools/probes/mx_probe*.hip (profiles/r02_mx_mfma_layout_probe_*.txt;
// no ISA document offline):
//   lane l = (i = l & 15, g = l >> 4) supplies row i of the 16 x 128 operand; its 8 VGPRs hold TWO runs of 16 bytes:
//   registers 0..3 = k in [16 g, 16 g + 16), registers 4..7 = k in [64 + 16 g, 64 + 16 g + 16) -- i.e. 16-byte chunks g and
//   4 + g of the 128-byte row, the same two chunks the bf16 kernel's two k steps read;
//   the scale VGPR of lane (i, q) is the E8M0 scale of row i, block q = k in [32 q, 32 q + 32) (byte 0 with op_sel 0);
//   D as for every 16x16 MFMA: lane l, register r -> D[4 (l >> 4) + r][l & 15].
// v_cvt_pk_fp8_f32 rounds to nearest even and returns NaN above 464, so values are clamped to +-448 before it.
//
// Data formats (MX, OCP): activations [M][Cp] e4m3 bytes + scales [M][Cp / 32] E8M0 bytes, Cp = C rounded up to 128 (pad
// channels are zero); weights Bt8[N][Kp] with k = (cs * T + tap) * 128 + ci (channel slice of 128 outer, taps inner) and
// scales Bs[N][Kp / 32].  A k tile = 128 k = one tap of one 128-channel slice = ONE MFMA per 16x16 output fragment.
//
// Kernel structure: k_gemm_bf16x.hip (256-row tiles, 8 waves, LDS-DMA with the XOR swizzle on the source side, two LDS
// stages, one barrier per k tile); additionally each stage carries the 4 scale bytes of every tile row, fetched by
// 4-byte LDS-DMA (one 64-row piece per wave), and read back one byte per fragment (ds_read_u8).
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_bf16_epi.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

__device__ __forceinline__ float qbf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float qbf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned qbf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned qpack_bf16x2(float a, float b) { return xpack_bf16x2(a, b); }   // one v_cvt_pk_bf16_f32 (k_gemm_bf16_epi.hpp)

// E8M0 scale byte of a block whose largest magnitude is amax: 2^(floor(log2 amax) - 8) (e4m3 emax = 8), so the scaled
// block lies in [256, 512) at its maximum and is clamped to 448 (the OCP MX rule).  Returns the byte and 1 / scale.
__device__ __forceinline__ unsigned mx_scale_byte(float amax, float* inv_scale) {
    const unsigned ex = (__float_as_uint(amax) >> 23) & 0xFFu;    // biased exponent = floor(log2 amax) + 127
    unsigned b = ex > 9u ? ex - 8u : 1u;                          // keep 1 <= byte (amax == 0 or denormal: everything quantises to 0)
    if (b > 253u) b = 253u;
    *inv_scale = __uint_as_float((254u - b) << 23);               // 2^(127 - b)
    return b;
}
__device__ __forceinline__ unsigned cvt4_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// =====================================================================================================================
// GEMM
// =====================================================================================================================
// One k tile of a wave: output column fragments [NB0, NB0 + CNT) of its tile against all MI row fragments.
// Register budget (256 per wave at two waves per SIMD): 4 MI NI accumulators + 8 per resident fragment.  With NI = 5 all
// five weight fragments + a double-buffered activation fragment do not fit next to 160 accumulators (the compiler then
// single-buffers and every 5 MFMAs wait for an LDS round trip: measured 27 % of the fp8 rate), so a 320-wide wave tile is
// walked in two column halves (3 + 2 fragments resident), re-reading the activation fragments once more -- LDS has the
// headroom (42 instead of 26 reads per 40 MFMAs).
template <int MI, int NI, int NB0, int CNT>
__device__ __forceinline__ void mx_columns(const unsigned char* stage, f32x4 (&acc)[MI][NI], int a_base, int b_base, int as_base, int bs_base,
                                           int fr_off0, int fr_off1) {
    auto frag = [&](int base, int row16) -> i32x8 {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(stage + base + row16 * 2048 + fr_off0);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(stage + base + row16 * 2048 + fr_off1);
        i32x8 f;
        f[0] = (int)lo[0]; f[1] = (int)lo[1]; f[2] = (int)lo[2]; f[3] = (int)lo[3];
        f[4] = (int)hi[0]; f[5] = (int)hi[1]; f[6] = (int)hi[2]; f[7] = (int)hi[3];
        return f;
    };
    i32x8 fb[CNT];
    int sb[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) {
        fb[j] = frag(b_base, NB0 + j);
        sb[j] = (int)stage[bs_base + (NB0 + j) * 64];
    }
    i32x8 fa[2];
    int sa[2];
    fa[0] = frag(a_base, 0);
    sa[0] = (int)stage[as_base];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (mi + 1 < MI) {   // the next row group's fragment is requested before this group's MFMAs issue
            fa[(mi + 1) & 1] = frag(a_base, mi + 1);
            sa[(mi + 1) & 1] = (int)stage[as_base + (mi + 1) * 64];
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j)
            acc[mi][NB0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[mi & 1], acc[mi][NB0 + j], 0, 0, 0, sb[j], 0, sa[mi & 1]);
    }
    // pin the issue order spelled out above (3 DS reads per fragment: two b128 + the scale byte)
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * CNT + 3, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (mi + 1 < MI) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, CNT, 0);
    }
}

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm_fp8x_kernel(const ConvGemm p) {
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM == 256 && BN % 64 == 0 && BN <= 320, "scale pieces: 4 for A, BN / 64 <= 5 for B");
    constexpr int NA = BM / 64;
    constexpr int NB = BN / 64;
    constexpr int SCALE_OFF = (BM + BN) * 128;          // scale words follow the two operand tiles
    constexpr int STAGE = (BM + BN) * 132;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int tm = gw.tm;
    const int tn = gw.tn;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const long long pix_bytes = (long long)p.a_ld;          // bytes between pixels of the fp8 activation (= Cp)
    const long long spix_bytes = (long long)p.a_ld >> 5;    // bytes between pixels of its scale tensor
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* ASbase = reinterpret_cast<const char*>(p.a_scale);
    const char* BSbase = reinterpret_cast<const char*>(p.b_scale);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ sub;

    // operand tiles: piece j of a wave = tile rows (wave + 8 j) * 8 + sub
    // (32-bit pixel indices / byte offsets: the launcher checks that both operands are < 4 GiB -- registers are tight)
    int a_iy0[NA], a_ix0[NA], a_nbpix[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave + 8 * j) * 8 + sub;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_nbpix[j] = nb * (p.Hs * p.Ws);
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);
        a_ix0[j] = ox * p.stride - p.pad;
    }
    unsigned b_off[NB];            // byte offset of this lane's chunk in row n of Bt8; ~0u: row beyond N
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + (wave + 8 * j) * 8 + sub;
        b_off[j] = (n < p.N) ? (unsigned)n * (unsigned)p.b_ld + (unsigned)chunk * 16u : ~0u;
    }
    // scale words (4 bytes per tile row per k tile): piece s = rows 64 s + lane; pieces 0..3 = operand A (waves 0..3),
    // 4..4 + NB - 1 = operand B (waves 4..7, and wave 0 once more for the fifth piece of a 320-wide tile)
    int s_iy0 = -(1 << 28), s_ix0 = 0, s_nbpix = 0;
    if (wave < NA) {
        const int m = m0 + wave * 64 + lane;
        if (m < p.M) {
            const int nb = m / HoWo;
            const int rem = m - nb * HoWo;
            const int oy = rem / p.Wo;
            s_nbpix = nb * (p.Hs * p.Ws);
            s_iy0 = oy * p.stride - p.pad;
            s_ix0 = (rem - oy * p.Wo) * p.stride - p.pad;
        }
    }
    unsigned sb_off0 = ~0u;   // B scale piece of waves 4..7 (byte offset of row n in Bs)
    unsigned sb_off1 = ~0u;   // fifth B scale piece (wave 0, NB == 5)
    if (wave >= 4 && wave - 4 < NB) {
        const int n = n0 + (wave - 4) * 64 + lane;
        if (n < p.N) sb_off0 = (unsigned)n * (unsigned)(p.b_ld >> 5);
    }
    if (NB > 4 && wave == 0) {
        const int n = n0 + 4 * 64 + lane;
        if (n < p.N) sb_off1 = (unsigned)n * (unsigned)(p.b_ld >> 5);
    }

    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;

    auto issue = [&](int buf) {
        unsigned char* stage = smem_q + buf * STAGE;
        const long long c0b = (long long)cs * 128;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int iy = a_iy0[j] + ky;
            const int ix = a_ix0[j] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const long long pix = a_nbpix[j] + ((iy >> p.ups) * p.Ws + (ix >> p.ups));
            const char* src = ok ? Abase + pix * pix_bytes + c0b + chunk * 16 : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        const unsigned k0b = (unsigned)kt_next * 128u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const char* src = b_off[j] != ~0u ? Bbase + (b_off[j] + k0b) : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + BM * 128 + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        if (wave < NA) {
            const int iy = s_iy0 + ky;
            const int ix = s_ix0 + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const long long pix = s_nbpix + ((iy >> p.ups) * p.Ws + (ix >> p.ups));
            const char* src = ok ? ASbase + pix * spix_bytes + cs * 4 : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + SCALE_OFF + wave * 256), 4, 0, 0);
        } else if (wave - 4 < NB) {
            const char* src = sb_off0 != ~0u ? BSbase + (sb_off0 + (unsigned)kt_next * 4u) : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + SCALE_OFF + BM * 4 + (wave - 4) * 256), 4, 0, 0);
        }
        if (NB > 4 && wave == 0) {
            const char* src = sb_off1 != ~0u ? BSbase + (sb_off1 + (unsigned)kt_next * 4u) : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + SCALE_OFF + BM * 4 + 4 * 256), 4, 0, 0);
        }
        const bool wrap_x = (kx + 1 == p.KW);
        const bool wrap_y = wrap_x && (ky + 1 == p.KH);
        kx = wrap_x ? 0 : kx + 1;
        ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
        cs = wrap_y ? cs + 1 : cs;
        ++kt_next;
    };

    // fragments: lane (c = lane & 15, g = lane >> 4) reads chunks g and 4 + g of row base + c (swizzled by c & 7) and the
    // scale byte g of that row
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    const int fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
    const int a_base = wm * 16 * MI * 128;
    const int b_base = BM * 128 + wn * 16 * NI * 128;
    const int as_base = SCALE_OFF + (wm * 16 * MI + c15) * 4 + g4;
    const int bs_base = SCALE_OFF + BM * 4 + (wn * 16 * NI + c15) * 4 + g4;

    f32x4 acc[MI][NI];
    issue(0);
    gemm_acc_init_bf16<MI, NI, WM, WN, 0>(p, acc, m0, n0, wave, lane, HoWo);   // zero, or the residual tile (ConvGemm::resid_acc), behind the first k tile's DMA
    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
        __syncthreads();                    // k tile t (operands and scales) is in LDS; every wave is done with stage cur ^ 1
        if (t + 1 < n_t) issue(cur ^ 1);
        const unsigned char* stage = smem_q + cur * STAGE;
        constexpr int NH = (NI > 4) ? 2 : 1;
        constexpr int NI0 = (NI + NH - 1) / NH;
        mx_columns<MI, NI, 0, NI0>(stage, acc, a_base, b_base, as_base, bs_base, fr_off0, fr_off1);
        if constexpr (NH == 2) mx_columns<MI, NI, NI0, NI - NI0>(stage, acc, a_base, b_base, as_base, bs_base, fr_off0, fr_off1);
    }

    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual -> bf16 (or fp32 slab / fp32 output): the LDS-transposed, row-coalesced store shared
    // with k_gemm_bf16x.hip (N % 8 == 0, ldc % 8 == 0 checked by the launcher; no GEGLU pairing on this kernel)
    gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_q, m0, n0, z, wave, lane, HoWo);
}

static const GemmTileInfo kTilesQ[kNumGemmTilesQ] = {{256, 320, "256x320q"}, {256, 256, "256x256q"}, {256, 128, "256x128q"}};
const GemmTileInfo& gemm_tile_info_q(int cfg) { return kTilesQ[cfg]; }

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg_fp8x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_fp8x_kernel<MI, NI, WM, WN>;
    constexpr size_t lds = 2 * (size_t)(16 * MI * WM + 16 * NI * WN) * 132;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_fp8x(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesQ) return hipErrorInvalidValue;
    if ((p.a_ld % 128) || (p.b_ld % 128) || !p.zero_page || !p.a_scale || !p.b_scale || p.geglu) return hipErrorInvalidValue;
    if ((p.N & 7) || (p.ldc & 7) || (p.resid && (p.ldr & 7))) return hipErrorInvalidValue;   // 16-byte epilogue only
    const int bm = kTilesQ[cfg].bm, bn = kTilesQ[cfg].bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bn - 1) / bn;
    const dim3 grid = gemm_grid(p, MT * NT);
    switch (cfg) {
        case 0: return launch_cfg_fp8x<8, 5, 2, 4>(p, grid, stream);
        case 1: return launch_cfg_fp8x<8, 4, 2, 4>(p, grid, stream);
        case 2: return launch_cfg_fp8x<4, 4, 4, 2>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

// =====================================================================================================================
// weight packing: fp32 OIHW -> e4m3 Bt8[N][Kp] (k = (cs * T + tap) * 128 + ci) + E8M0 scales Bs[N][Kp / 32]
// =====================================================================================================================
// one thread per (n, slot = (cs, tap), 32-channel block)
__global__ void pack_conv_weight_fp8_kernel(const float* __restrict__ w, unsigned char* __restrict__ bt, unsigned char* __restrict__ bs,
                                            int cout, int cin, int kh, int kw, int cp) {
    const int T = kh * kw;
    const int nblk = cp / 32;                      // blocks per tap
    const long long total = (long long)cout * T * nblk;
    const long long Kp = (long long)cp * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int blk = (int)(i % nblk);           // 32-channel block of the padded channel axis
        const long long r = i / nblk;
        const int tap = (int)(r % T);
        const int n = (int)(r / T);
        const int c0 = blk * 32;
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = c0 + j;
            v[j] = c < cin ? w[((long long)n * cin + c) * T + tap] : 0.f;
            amax = fmaxf(amax, fabsf(v[j]));
        }
        float inv;
        const unsigned sb = mx_scale_byte(amax, &inv);
        const int cs = c0 / 128, ci0 = c0 - cs * 128;
        const long long k0 = ((long long)cs * T + tap) * 128 + ci0;
        unsigned* dst = reinterpret_cast<unsigned*>(bt + (long long)n * Kp + k0);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = cvt4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
        bs[(long long)n * (Kp / 32) + k0 / 32] = (unsigned char)sb;
    }
}

hipError_t launch_pack_conv_weight_fp8(const float* w_oihw, void* bt8, void* bs, int cout, int cin, int kh, int kw, hipStream_t s) {
    const int cp = (cin + 127) / 128 * 128;
    const long long total = (long long)cout * kh * kw * (cp / 32);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv_weight_fp8_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, reinterpret_cast<unsigned char*>(bt8),
                       reinterpret_cast<unsigned char*>(bs), cout, cin, kh, kw, cp);
    return hipGetLastError();
}

// =====================================================================================================================
// GroupNorm(+SiLU) apply with MXFP8 output: the quantisation of the conv's input is fused into the normalisation that
// produces it (statistics: gn_stats_bf16_kernel, unchanged).  x bf16 [n][hw][ldx] -> y e4m3 [n][hw][Cp] + scales [n][hw][Cp/32]
// =====================================================================================================================
struct Q8 { float v[8]; };
__device__ __forceinline__ Q8 qunpack8(u32x4 w) {
    Q8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[2 * i] = __uint_as_float(w[i] << 16); r.v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    return r;
}

template <bool SILU>
__global__ void gn_apply_fp8_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ y, unsigned char* __restrict__ ys,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int hw, int C, int ldx, int Cp, int G,
                                    float eps, int stat_chunks, int stat_rows, const double* __restrict__ part, int rows_per_chunk) {
    __shared__ float s_mean_hi[64], s_mean_lo[64], s_rstd[64];
    __shared__ double s_red[3][64][8];
    const int cq = C >> 3;
    const int R = blockDim.x / cq;
    const int tid = threadIdx.x;
    const int smp = blockIdx.y;
    const int cpg = C / G;
    const int c8 = tid % cq;
    const int r0 = tid / cq;
    // the thread's first row is requested before the statistics are finalised (as in gn_apply_bf16_kernel)
    const int row_first = blockIdx.x * rows_per_chunk + r0;
    u32x4 w_first = {0u, 0u, 0u, 0u};
    if (row_first < min(blockIdx.x * rows_per_chunk + rows_per_chunk, hw)) w_first = *reinterpret_cast<const u32x4*>(x + (long long)smp * hw * ldx + c8 * 8 + (long long)row_first * ldx);
    float gm[8], bt[8], mean_hi[8], mean_lo[8], rstd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {     // (round 6: gamma / beta requested in front of gn_finalize's barriers, not behind them)
        gm[i] = gamma[c8 * 8 + i];
        bt[i] = beta[c8 * 8 + i];
    }
    gn_finalize(part, smp, G, cpg, hw, stat_chunks, stat_rows, eps, s_red, s_mean_hi, s_mean_lo, s_rstd);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = c8 * 8 + i;
        mean_hi[i] = s_mean_hi[ch / cpg];
        mean_lo[i] = s_mean_lo[ch / cpg];
        rstd[i] = s_rstd[ch / cpg];
    }
    const int row_begin = blockIdx.x * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, hw);
    const long long xbase = (long long)smp * hw * ldx + c8 * 8;
    const long long ybase = (long long)smp * hw * Cp;
    const long long sbase = (long long)smp * hw * (Cp >> 5);
    const int pad8 = (Cp - C) >> 3;    // 8-channel groups of zero padding per row (C = 320: 8)
    auto emit = [&](const int row, const u32x4 w) {
        Q8 v = qunpack8(w);
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = ((v.v[i] - mean_hi[i]) - mean_lo[i]) * rstd[i];
            t = t * gm[i] + bt[i];
            if (SILU) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));   // (hardware reciprocal, 1 ulp: at the batches of configs[2..4] the IEEE division made this pass VALU-bound)
            v.v[i] = t;
            amax = fmaxf(amax, fabsf(t));
        }
        // a 32-channel MX block = 4 consecutive threads (cq % 4 == 0, so they share a row and sit in one aligned lane quad)
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        float inv;
        const unsigned sb = mx_scale_byte(amax, &inv);
        u32x2 o;
        o[0] = cvt4_e4m3(v.v[0] * inv, v.v[1] * inv, v.v[2] * inv, v.v[3] * inv);
        o[1] = cvt4_e4m3(v.v[4] * inv, v.v[5] * inv, v.v[6] * inv, v.v[7] * inv);
        *reinterpret_cast<u32x2*>(y + ybase + (long long)row * Cp + c8 * 8) = o;
        if ((c8 & 3) == 0) ys[sbase + (long long)row * (Cp >> 5) + (c8 >> 2)] = (unsigned char)sb;
        for (int g = c8; g < pad8; g += cq) {     // (C = 32: four threads per row write twelve pad groups)
            *reinterpret_cast<u32x2*>(y + ybase + (long long)row * Cp + C + g * 8) = u32x2{0u, 0u};
            if ((g & 3) == 0) ys[sbase + (long long)row * (Cp >> 5) + ((C + g * 8) >> 5)] = (unsigned char)127;
        }
    };
    // one row ahead: the next row's load is in flight while this row is normalised and quantised (the first was requested in front of gn_finalize)
    if (row_first < row_end) {
        int row = row_first;
        u32x4 w = w_first;
        for (;;) {
            const int nxt = row + R;
            u32x4 wn = w;
            if (nxt < row_end) wn = *reinterpret_cast<const u32x4*>(x + xbase + (long long)nxt * ldx);
            emit(row, w);
            if (nxt >= row_end) break;
            row = nxt;
            w = wn;
        }
    }
}

hipError_t launch_group_norm_fp8(const void* x, void* y8, void* y_scale, const float* gamma, const float* beta, int n, int hw, int c,
                                 int ldx, int n_group, float eps, bool silu, void* partials, hipStream_t stream, const GnTune* tp) {
    if ((c & 31) || (ldx & 7) || ldx < c || n_group > 64 || c % n_group || c / 8 > 1024) return hipErrorInvalidValue;
    const GnTune t = tp ? *tp : GnTune();
    hipError_t e = launch_group_norm_bf16_stats(x, n, hw, c, ldx, n_group, partials, stream, t);
    if (e != hipSuccess) return e;
    const GnGeomH g = gn_geom_bf16(n, hw, c, t);    // the geometry of the bf16 GroupNorm (k_bf16.hip): the statistics pass is launch_group_norm_bf16's first half
    const int cp = (c + 127) / 128 * 128;
    auto xs = reinterpret_cast<const unsigned short*>(x);
    auto yq = reinterpret_cast<unsigned char*>(y8);
    auto ysc = reinterpret_cast<unsigned char*>(y_scale);
    const double* part = reinterpret_cast<const double*>(partials);
    if (silu)
        hipLaunchKernelGGL(gn_apply_fp8_kernel<true>, dim3(g.chunks, n), dim3(g.threads), 0, stream, xs, yq, ysc, gamma, beta, hw, c, ldx, cp,
                           n_group, eps, g.chunks, g.rows_per_chunk, part, g.rows_per_chunk);
    else
        hipLaunchKernelGGL(gn_apply_fp8_kernel<false>, dim3(g.chunks, n), dim3(g.threads), 0, stream, xs, yq, ysc, gamma, beta, hw, c, ldx, cp,
                           n_group, eps, g.chunks, g.rows_per_chunk, part, g.rows_per_chunk);
    return hipGetLastError();
}

// =====================================================================================================================
// precision = 2 beyond the ResBlock convolutions (option fp8_linear): the transformer blocks' Linear layers and the 1x1 / up / down
// convolutions take MXFP8 operands too (BASELINE.json configs[4]: "fp8 conv+attn"; reference layers unet/mod.rs:397,425,468,479,
// 553,580,645-651 and autoencoder/mod.rs:319,568-604).  Their inputs are quantised by the kernel that produces them where that is a
// row-wise kernel -- LayerNorm, the GEGLU gate, GroupNorm (above) -- and by quantize_bf16_fp8_kernel otherwise (attention outputs,
// the residual stream in front of proj_out / skip / up / down convolutions).
// =====================================================================================================================
// 8 bf16 -> the thread's 8 e4m3 bytes + (one thread in four) the block's scale byte.  The four threads of a 32-channel block are an
// aligned lane quad.  Pad channels [C, Cp) are written as zeros with scale byte 127 by the threads `c8 < pad8`.
__device__ __forceinline__ void mx_store8(const Q8& v, unsigned char* yrow, unsigned char* srow, int c8, int C, int Cp) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v.v[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    float inv;
    const unsigned sb = mx_scale_byte(amax, &inv);
    u32x2 o;
    o[0] = cvt4_e4m3(v.v[0] * inv, v.v[1] * inv, v.v[2] * inv, v.v[3] * inv);
    o[1] = cvt4_e4m3(v.v[4] * inv, v.v[5] * inv, v.v[6] * inv, v.v[7] * inv);
    *reinterpret_cast<u32x2*>(yrow + c8 * 8) = o;
    if ((c8 & 3) == 0) srow[c8 >> 2] = (unsigned char)sb;
    const int pad8 = (Cp - C) >> 3;
    for (int g = c8; g < pad8; g += (C >> 3)) {     // every pad group is written, also when the row has fewer threads than pad groups (C = 32)
        *reinterpret_cast<u32x2*>(yrow + C + g * 8) = u32x2{0u, 0u};
        if ((g & 3) == 0) srow[(C + g * 8) >> 5] = (unsigned char)127;
    }
}

// bf16 [rows][ldx] (C channels, C % 32 == 0) -> MXFP8 [rows][Cp] + scales [rows][Cp / 32]; one thread per 8 channels
__global__ void quantize_bf16_fp8_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ y, unsigned char* __restrict__ ys,
                                         long long rows, int C, int ldx, int Cp) {
    const int cq = C >> 3;
    const long long total = rows * cq;
    const long long rounded = (total + 3) / 4 * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += (long long)gridDim.x * blockDim.x) {
        const long long r = (i < total ? i : total - 1) / cq;
        const int c8 = (int)((i < total ? i : total - 1) - r * cq);
        const Q8 v = qunpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c8 * 8));
        mx_store8(v, y + r * Cp, ys + r * (Cp >> 5), c8, C, Cp);       // (cq % 4 == 0: a quad never straddles rows or the end)
    }
}
hipError_t launch_quantize_bf16_fp8(const void* x, void* q, void* s, long long rows, int c, int ldx, hipStream_t stream) {
    if ((c & 31) || (ldx & 7) || ldx < c) return hipErrorInvalidValue;
    const int cp = (c + 127) / 128 * 128;
    long long blocks = (rows * (c / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(quantize_bf16_fp8_kernel, dim3((int)blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(x),
                       reinterpret_cast<unsigned char*>(q), reinterpret_cast<unsigned char*>(s), rows, c, ldx, cp);
    return hipGetLastError();
}

// LayerNorm (unet/mod.rs:523-525) of a bf16 tensor with MXFP8 output: the row kernel of k_bf16.hip (L lanes per row, the row in
// registers, exact two-pass statistics) with the quantiser in its tail.  Lane l holds the 8-channel pieces f = l + i L: four consecutive
// lanes hold one 32-channel block (L % 4 == 0).
constexpr int kLnMaxVecQ = 4;
template <int L>
__global__ __launch_bounds__(256) void layer_norm_fp8_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ y, unsigned char* __restrict__ ys,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int rows, int C, int Cp, float eps) {
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int sub = lane / L, l = lane % L;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool row_ok = row < rows;
    const int cq = C >> 3;
    const unsigned short* xr = x + (long long)(row_ok ? row : 0) * C;
    Q8 v[kLnMaxVecQ];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxVecQ; ++i) {
        const int f = l + i * L;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i].v[j] = 0.f;
        if (f < cq && row_ok) {
            v[i] = qunpack8(*reinterpret_cast<const u32x4*>(xr + f * 8));
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i].v[j];
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxVecQ; ++i) {
        const int f = l + i * L;
        if (f < cq) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i].v[j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    const long long rr = row_ok ? row : 0;
#pragma unroll
    for (int i = 0; i < kLnMaxVecQ; ++i) {      // uniform trip count: the quad shuffles of mx_store8 need all four lanes of a block
        const int f = l + i * L;
        const int ff = f < cq ? f : 0;
        Q8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = (v[i].v[j] - mean) * rstd * gamma[ff * 8 + j] + beta[ff * 8 + j];
        if ((i * L) < cq) {                     // uniform over the L lanes: cq % L need not be 0, but cq % 4 == 0 keeps quads whole
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(o.v[j]));
            amax = fmaxf(amax, __shfl_xor(amax, 1));
            amax = fmaxf(amax, __shfl_xor(amax, 2));
            if (f < cq && row_ok) {
                float inv;
                const unsigned sb = mx_scale_byte(amax, &inv);
                u32x2 w;
                w[0] = cvt4_e4m3(o.v[0] * inv, o.v[1] * inv, o.v[2] * inv, o.v[3] * inv);
                w[1] = cvt4_e4m3(o.v[4] * inv, o.v[5] * inv, o.v[6] * inv, o.v[7] * inv);
                unsigned char* yrow = y + rr * Cp;
                unsigned char* srow = ys + rr * (Cp >> 5);
                *reinterpret_cast<u32x2*>(yrow + f * 8) = w;
                if ((f & 3) == 0) srow[f >> 2] = (unsigned char)sb;
                const int pad8 = (Cp - C) >> 3;
                for (int g = f; g < pad8; g += cq) {
                    *reinterpret_cast<u32x2*>(yrow + C + g * 8) = u32x2{0u, 0u};
                    if ((g & 3) == 0) srow[(C + g * 8) >> 5] = (unsigned char)127;
                }
            }
        }
    }
}
hipError_t launch_layer_norm_fp8(const void* x, void* y8, void* y_scale, const float* gamma, const float* beta, int rows, int c, float eps,
                                 hipStream_t stream) {
    if ((c & 31) || c > kLnMaxVecQ * 512) return hipErrorInvalidValue;
    const int cq = c >> 3, cp = (c + 127) / 128 * 128;
    auto xs = reinterpret_cast<const unsigned short*>(x);
    auto yq = reinterpret_cast<unsigned char*>(y8);
    auto ysc = reinterpret_cast<unsigned char*>(y_scale);
    if (cq <= 16 * kLnMaxVecQ)
        hipLaunchKernelGGL(layer_norm_fp8_kernel<16>, dim3((rows + 15) / 16), dim3(256), 0, stream, xs, yq, ysc, gamma, beta, rows, c, cp, eps);
    else if (cq <= 32 * kLnMaxVecQ)
        hipLaunchKernelGGL(layer_norm_fp8_kernel<32>, dim3((rows + 7) / 8), dim3(256), 0, stream, xs, yq, ysc, gamma, beta, rows, c, cp, eps);
    else
        hipLaunchKernelGGL(layer_norm_fp8_kernel<64>, dim3((rows + 3) / 4), dim3(256), 0, stream, xs, yq, ysc, gamma, beta, rows, c, cp, eps);
    return hipGetLastError();
}

// the GEGLU gate (unet/mod.rs:579-591) with MXFP8 output: proj bf16 [rows][2 H] -> out e4m3 [rows][Hp] + scales
__device__ __forceinline__ float gelu_erf_q(float x) { return gelu_gate_fast(x); }   // (k_common.hpp)
__global__ void geglu_fp8_kernel(const unsigned short* __restrict__ proj, unsigned char* __restrict__ y, unsigned char* __restrict__ ys, long long rows,
                                 int H, int Hp) {
    const int hq = H >> 3;
    const long long total = rows * hq;
    const long long rounded = (total + 3) / 4 * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += (long long)gridDim.x * blockDim.x) {
        const long long ii = i < total ? i : total - 1;
        const long long r = ii / hq;
        const int c8 = (int)(ii - r * hq);
        const Q8 a = qunpack8(*reinterpret_cast<const u32x4*>(proj + r * 2 * H + c8 * 8));
        const Q8 g = qunpack8(*reinterpret_cast<const u32x4*>(proj + r * 2 * H + H + c8 * 8));
        Q8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = a.v[j] * gelu_erf_q(g.v[j]);
        // (the bf16 gate kernel rounds its output to bf16 before the next GEMM reads it; the same rounding here keeps the two forms comparable)
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = __uint_as_float(qbf16_bits(o.v[j]) << 16);
        mx_store8(o, y + r * Hp, ys + r * (Hp >> 5), c8, H, Hp);
    }
}
hipError_t launch_geglu_fp8(const void* proj, void* y8, void* y_scale, long long rows, int hidden, hipStream_t stream) {
    if (hidden & 31) return hipErrorInvalidValue;
    const int hp = (hidden + 127) / 128 * 128;
    long long blocks = (rows * (hidden / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(geglu_fp8_kernel, dim3((int)blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned short*>(proj),
                       reinterpret_cast<unsigned char*>(y8), reinterpret_cast<unsigned char*>(y_scale), rows, hidden, hp);
    return hipGetLastError();
}

// Linear weight fp32 [in][out] (the reference's layout, python/save.py:19) -> e4m3 Bt8[out][Kp] + scales, Kp = roundup(in, 128):
// the k order of a 1x1 convolution (k = channel), so conv_gemm_fp8x_kernel reads it as one
__global__ void pack_linear_weight_fp8_kernel(const float* __restrict__ w, unsigned char* __restrict__ bt, unsigned char* __restrict__ bs, int cin, int cout, int kp) {
    const int nblk = kp / 32;
    const long long total = (long long)cout * nblk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int blk = (int)(i % nblk);
        const int n = (int)(i / nblk);
        const int c0 = blk * 32;
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int c = c0 + j;
            v[j] = c < cin ? w[(long long)c * cout + n] : 0.f;
            amax = fmaxf(amax, fabsf(v[j]));
        }
        float inv;
        const unsigned sb = mx_scale_byte(amax, &inv);
        unsigned* dst = reinterpret_cast<unsigned*>(bt + (long long)n * kp + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = cvt4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
        bs[(long long)n * nblk + blk] = (unsigned char)sb;
    }
}
hipError_t launch_pack_linear_weight_fp8(const float* w_in_out, void* bt8, void* bs, int cin, int cout, hipStream_t s) {
    const int kp = (cin + 127) / 128 * 128;
    const long long total = (long long)cout * (kp / 32);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_linear_weight_fp8_kernel, dim3(blocks), dim3(256), 0, s, w_in_out, reinterpret_cast<unsigned char*>(bt8),
                       reinterpret_cast<unsigned char*>(bs), cin, cout, kp);
    return hipGetLastError();
}

// plain quantiser (operator-level entry point / tests): fp32 [rows][C] -> e4m3 [rows][Cp] + scales [rows][Cp / 32];
// one thread per (row, 32-channel block) of the padded row
__global__ void quantize_fp8_kernel(const float* __restrict__ x, unsigned char* __restrict__ q, unsigned char* __restrict__ s, long long rows,
                                    int C, int Cp) {
    const int nblk = Cp >> 5;
    const long long total = rows * nblk;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / nblk;
        const int c0 = (int)(i - r * nblk) * 32;
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            v[j] = (c0 + j < C) ? x[r * C + c0 + j] : 0.f;
            amax = fmaxf(amax, fabsf(v[j]));
        }
        float inv;
        const unsigned sb = mx_scale_byte(amax, &inv);
        unsigned* dst = reinterpret_cast<unsigned*>(q + r * Cp + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = cvt4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
        s[r * nblk + (c0 >> 5)] = (unsigned char)(c0 < C ? sb : 127u);
    }
}

hipError_t launch_quantize_fp8(const float* x, void* q, void* s, long long rows, int c, hipStream_t stream) {
    const int cp = (c + 127) / 128 * 128;
    long long blocks = (rows * (cp / 32) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3((int)blocks), dim3(256), 0, stream, x, reinterpret_cast<unsigned char*>(q),
                       reinterpret_cast<unsigned char*>(s), rows, c, cp);
    return hipGetLastError();
}

// dequantise (tests / operator-level entry point): e4m3 [rows][Cp] + scales -> fp32 [rows][C]
__global__ void dequant_fp8_kernel(const unsigned char* __restrict__ q, const unsigned char* __restrict__ s, float* __restrict__ out,
                                   long long rows, int C, int Cp) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const unsigned char v = q[r * Cp + c];
        const int e = (v >> 3) & 15, m = v & 7;
        float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m * 0.125f, e - 7);
        if (v & 0x80) f = -f;
        out[i] = f * ldexpf(1.0f, (int)s[r * (Cp >> 5) + (c >> 5)] - 127);
    }
}

hipError_t launch_dequant_fp8(const void* q, const void* s, float* out, long long rows, int c, hipStream_t stream) {
    const int cp = (c + 127) / 128 * 128;
    long long blocks = (rows * c + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dequant_fp8_kernel, dim3((int)blocks), dim3(256), 0, stream, reinterpret_cast<const unsigned char*>(q),
                       reinterpret_cast<const unsigned char*>(s), out, rows, c, cp);
    return hipGetLastError();
}

}  // namespace sdmi
