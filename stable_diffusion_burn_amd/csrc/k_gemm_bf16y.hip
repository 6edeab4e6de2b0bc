// k_gemm_bf16y.hip -- large-tile bf16 implicit-GEMM conv / linear on v_mfma_f32_32x32x16_bf16 (precision = 1).  EXPERIMENTAL (tiles 300 + x,
// chosen only by option gemm_tile / a tuning-table entry): written at the end of round 2, compiled and index-checked on the CPU
// (tests/test_gemm3y_layout_cpu.py), NOT yet run on an MI355X.
//
// Why: the same finding as for k_gemm3y.hip (profiles/README.md, "Where the k loop's time goes").  k_gemm_bf16x.hip keeps the matrix pipe
// about 50 % busy in every form of its k loop, its bytes come from L2 at a fifth of the rate a CU can land them, and what is left is the
// number of instructions: 80 v_mfma_f32_16x16x32_bf16 per wave per 64-deep k tile with 26 fragment reads and the DMA's address arithmetic
// between them.  v_mfma_f32_32x32x16_bf16 does the same flops in 40 matrix instructions (and is the form that reaches the full bf16 rate):
// 256 x 320 workgroup tile as 4 x 2 waves of 64 pixels x 160 channels = 2 x 5 fragments of 32 x 32, 160 accumulator registers as before.
//
// Operands in HBM exactly as k_gemm_bf16x.hip reads them (bf16 NHWC activations, weights [N][K] in the kernels' k order); a k tile is 64
// channels = 128 bytes per row = four 16-deep matrix steps s; lane (c = lane & 31, hi = lane >> 5) supplies row / column c and the 16-byte chunk
// 2 s + hi of its row on BOTH sides (k = 8 (2 s + hi) .. + 7), so nothing about the packing changes.  LDS: both operand tiles as 128-byte rows,
// chunk c8 of tile row r in slot c8 ^ ((r >> 1) & 7) (swizzle applied by the DMA on the source address): conflict-free ds_read_b128 over
// 32 consecutive rows (k_gemm3y.hip has the argument).  D: lane holds pixel c and channels (r & 3) + 8 (r >> 2) + 4 hi of a fragment in acc[r].
// Swapped operands as everywhere: A operand = weights, B operand = activations.  First version: the plain loop structure (two LDS stages,
// one __syncthreads() per k tile, straight-line DMA issue, hipcc's waits), bias + time-embedding row + bf16 residual epilogue, bf16 or fp32
// output, split-K slabs for the separate reduce kernel; no GEGLU mode, no in-launch combine.
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

__device__ __forceinline__ float ybf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float ybf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned yf32_to_bf16_bits(float f) {      // round to nearest even (as k_gemm_bf16x.hip)
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned ypack_bf16x2(float a, float b) { return yf32_to_bf16_bits(a) | (yf32_to_bf16_bits(b) << 16); }

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm_bf16y_kernel(const ConvGemm p) {
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "every wave issues whole 8-row DMA pieces");
    constexpr int NA = BM / 64;               // activation pieces (8 rows x 128 B) per wave per k tile
    constexpr int NB = BN / 64;               // weight pieces per wave
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_by[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int m0 = gw.tm * BM;
    const int n0 = gw.tn * BN;
    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const unsigned pix_bytes = (unsigned)p.a_ld * 2u;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // DMA piece j of a wave covers rows (wave + 8 j) * 8 + sub of an operand tile, sub = lane >> 3; LDS slot lane & 7 receives global chunk
    // (lane & 7) ^ ((row >> 1) & 7), and (row >> 1) & 7 = ((wave & 1) * 4 + (sub >> 1)) & 7 whatever j is -- the same for both operand tiles
    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ ((((wave & 1) << 2) + (sub >> 1)) & 7);
    int a_iy0[NA], a_ix0[NA];
    unsigned a_off[NA];               // (operands are < 4 GiB: launch_gemm checks)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave + 8 * j) * 8 + sub;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_off[j] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * pix_bytes + chunk * 16;
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
        a_ix0[j] = ox * p.stride - p.pad;
    }
    unsigned b_off[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int n = n0 + (wave + 8 * j) * 8 + sub;
        if (n >= p.N) n = p.N - 1;      // rows past N (ragged last tile) fetch the last valid row: real memory, never stored
        b_off[j] = (unsigned)n * (unsigned)p.b_ld * 2u + chunk * 16;
    }
    int cs = kt_begin / T;
    const int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;

    // DMA of k tile kt_next into `stage`, straight-line; after the last tile the same tile is fetched once more into the dead stage
    auto issue = [&](unsigned char* stage) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int iy = a_iy0[j] + ky;
            const int ix = a_ix0[j] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const unsigned off = a_off[j] + (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups)) * pix_bytes + (unsigned)cs * 128u;
            const char* src = (ok ? Abase : zero) + (ok ? off : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const char* src = Bbase + (b_off[j] + (unsigned)kt_next * 128u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
        }
        const bool adv = kt_next + 1 < kt_end;
        const bool wrap_x = (kx + 1 == p.KW);
        const bool wrap_y = wrap_x && (ky + 1 == p.KH);
        const int kx1 = wrap_x ? 0 : kx + 1;
        const int ky1 = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
        const int cs1 = wrap_y ? cs + 1 : cs;
        kx = adv ? kx1 : kx;
        ky = adv ? ky1 : ky;
        cs = adv ? cs1 : cs;
        kt_next = adv ? kt_next + 1 : kt_next;
    };

    // fragment addresses: tile row r of an operand, chunk 2 s + hi, slot (chunk) ^ ((r >> 1) & 7).  The fragments of a lane are 32 rows apart, so
    // (r >> 1) & 7 = (c >> 1) & 7 for all of them: one swizzled chunk offset per matrix step, the fragment index goes into the read's immediate
    const int c = lane & 31, hi = lane >> 5;
    const int sw = (c >> 1) & 7;
    const int a_lane = (wm * MI * 32 + c) * 128;                  // + mi * 4096
    const int b_lane = A_BYTES + (wn * NI * 32 + c) * 128;        // + ni * 4096
    int g_off[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) g_off[s_] = ((2 * s_ + hi) ^ sw) << 4;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    issue(smem_by);
    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        issue(smem_by + (cur ^ 1) * STAGE);
        const unsigned char* stage = smem_by + cur * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 fa[MI], fb[NI];
            const unsigned char* pb = stage + b_lane + g_off[s];
            const unsigned char* pa = stage + a_lane + g_off[s];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(pb + ni * 4096);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(pa + mi * 4096);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]), acc[mi][ni], 0, 0, 0);
        }
    }
    // the last k tile was fetched twice; that copy must have landed before the epilogue reuses the stages (and before the wave ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual, then bf16 or fp32 store / split-K slab --------------------------------
    // every wave transposes one 32 x 32 fragment at a time through its own LDS scratch and writes whole row segments with 16-byte lanes
    const bool split = p.splits > 1;
    const bool out_f32 = split || p.out_mode == 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = !split && p.resid;
    const bool vec_ok = ((p.N & 7) == 0) && ((ldc & 7) == 0) && ((p.ldr & 7) == 0 || !has_resid);
    constexpr int LDSW = 36;                // scratch row stride in floats
    if (vec_ok) {
        __syncthreads();                    // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_by + wave * (32 * LDSW * 4));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mrow0 = m0 + (wm * MI + mi) * 32;
            const int m_lane = mrow0 + c;
            const int smp = (m_lane < p.M ? m_lane : 0) / HoWo;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int nf0 = n0 + (wn * NI + ni) * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nf0 + 8 * q + 4 * hi;
                    f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                    if (!split && n < p.N) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    }
                    *reinterpret_cast<f32x4*>(scr + c * LDSW + 8 * q + 4 * hi) = v;
                }
                __builtin_amdgcn_wave_barrier();
                if (!out_f32) {
                    // 32 rows x 4 chunks of 8 bf16
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 16 + (lane >> 2), c8 = lane & 3;
                        const int m = mrow0 + row, n = nf0 + c8 * 8;
                        if (m < p.M && n < p.N) {
                            f32x4 lo = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8);
                            f32x4 hv = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8 + 4);
                            if (has_resid) {
                                const u32x4 r = *reinterpret_cast<const u32x4*>(Rh + (long long)m * p.ldr + n);
                                lo[0] += ybf16_lo(r[0]); lo[1] += ybf16_hi(r[0]); lo[2] += ybf16_lo(r[1]); lo[3] += ybf16_hi(r[1]);
                                hv[0] += ybf16_lo(r[2]); hv[1] += ybf16_hi(r[2]); hv[2] += ybf16_lo(r[3]); hv[3] += ybf16_hi(r[3]);
                            }
                            const u32x4 o = {ypack_bf16x2(lo[0], lo[1]), ypack_bf16x2(lo[2], lo[3]), ypack_bf16x2(hv[0], hv[1]), ypack_bf16x2(hv[2], hv[3])};
                            *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o;
                        }
                    }
                } else {
                    // 32 rows x 8 chunks of 4 floats
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + (lane >> 3), c4 = lane & 7;
                        const int m = mrow0 + row, n = nf0 + c4 * 4;
                        if (m < p.M && n < p.N) {
                            f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                            if (has_resid) {
                                const u32x2 r = *reinterpret_cast<const u32x2*>(Rh + (long long)m * p.ldr + n);
                                v[0] += ybf16_lo(r[0]); v[1] += ybf16_hi(r[0]); v[2] += ybf16_lo(r[1]); v[3] += ybf16_hi(r[1]);
                            }
                            *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        // odd strides / N not a multiple of 8: element-wise stores straight from the accumulators
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + (wm * MI + mi) * 32 + c;
            if (m >= p.M) continue;
            const int smp = m / HoWo;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + (wn * NI + ni) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (n < p.N) {
                        float sv = acc[mi][ni][r];
                        if (!split) {
                            if (p.bias) sv += p.bias[n];
                            if (p.rowvec) sv += p.rowvec[(long long)smp * p.rowvec_stride + n];
                            if (p.resid) sv += __uint_as_float((unsigned)Rh[(long long)m * p.ldr + n] << 16);
                        }
                        if (out_f32) Cf[(long long)m * ldc + n] = sv;
                        else Ch[(long long)m * ldc + n] = (unsigned short)yf32_to_bf16_bits(sv);
                    }
                }
        }
    }
}

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg_bf16y(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    static bool attr_set = false;
    auto k = conv_gemm_bf16y_kernel<MI, NI, WM, WN>;
    constexpr size_t lds = 2 * (size_t)(32 * MI * WM + 32 * NI * WN) * 128;
    static_assert(lds <= 160 * 1024 && 8 * 32 * 36 * 4 <= lds, "stages / epilogue scratch");
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

// tile_cfg as gemm_tile_info_x(): 0 = 256x320, 1 = 256x256, 2 = 256x128, 3 = 128x320
hipError_t launch_conv_gemm_bf16y(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesX) return hipErrorInvalidValue;
    if ((p.Cin % 64) || !p.zero_page || p.geglu || p.counters) return hipErrorInvalidValue;
    const int bm = gemm_tile_info_x(cfg).bm, bn = gemm_tile_info_x(cfg).bn;
    const int tiles = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    const dim3 grid = gemm_grid(p, tiles);
    switch (cfg) {
        case 0: return launch_cfg_bf16y<2, 5, 4, 2>(p, grid, stream);   // 256 x 320: waves of 64 x 160
        case 1: return launch_cfg_bf16y<2, 4, 4, 2>(p, grid, stream);   // 256 x 256: 64 x 128
        case 2: return launch_cfg_bf16y<2, 2, 4, 2>(p, grid, stream);   // 256 x 128: 64 x 64
        case 3: return launch_cfg_bf16y<1, 5, 4, 2>(p, grid, stream);   // 128 x 320: 32 x 160
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
