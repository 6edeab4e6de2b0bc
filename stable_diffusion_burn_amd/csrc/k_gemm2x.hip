// k_gemm2x.hip -- large-tile fp32 implicit-GEMM conv / linear: the 8-wave, LDS-DMA staged structure of
// k_gemm_bf16x.hip with fp32 storage and v_mfma_f32_16x16x4_f32 (exact fp32 products and sums).
//
// A k tile is 32 floats = the same 128 bytes per row as 64 bf16, so the DMA addressing, the source-side XOR swizzle
// and the fragment addressing are byte-for-byte those of the bf16 kernel; a 16-byte fragment holds 4 consecutive
// k = four k-steps of the 16x16x4 MFMA (the k permutation of k_gemm2.hip).  Why it exists: the 4-wave kernel
// (k_gemm2.hip) stages through registers (global load -> VGPR -> ds_write) and keeps one or two waves per SIMD busy
// 62-81 % of the time; here a wave issues 9 DMA instructions and 26 fragment reads per 320 MFMAs (10 240 matrix-pipe
// cycles), nothing else, and the output tile leaves through the LDS-transposed coalesced epilogue.
// k order, weight packing, XCD-aware tile map, deterministic split-K slabs and the fused epilogue (bias +
// time-embedding row + residual) are those of k_gemm2.hip, so both kernels read the same packed weights.
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_epi.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

// tile list shared with the bf16 kernel: gemm_tile_info_x() (k_gemm_bf16x.hip)

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm2x_kernel(const ConvGemm p) {
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "every wave issues whole 8-row DMA pieces");
    constexpr int NA = BM / 64;               // A pieces (8 rows x 128 B) per wave per k tile
    constexpr int NB = BN / 64;               // B pieces per wave
    constexpr int STAGE = (BM + BN) * 128;    // bytes of one LDS stage

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x32[];

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
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const long long pix_bytes = (long long)p.a_ld * 4;
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
        b_src[j] = (n < p.N) ? Bbase + wrow * p.b_ld * 4 + chunk * 16 : nullptr;
    }

    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;

    auto issue = [&](int buf) {   // DMA of k tile kt_next into LDS stage buf; advances (cs, ky, kx)
        unsigned char* stage = smem_x32 + buf * STAGE;
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
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    const int fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
    const int a_base = wm * 16 * MI * 128;
    const int b_base = BM * 128 + wn * 16 * NI * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        if (t + 1 < n_t) issue(cur ^ 1);
        const unsigned char* stage = smem_x32 + cur * STAGE;
        // Fragment reads run one group of rows ahead of the MFMAs that use them: the reads of rows [g+1] are issued
        // before the MFMAs of rows [g], so the compiler's counted lgkmcnt waits find the data already there
        // (one ds_read_b128 per 4 NI MFMAs).
        constexpr int GM = (MI >= 8) ? 4 : 2;    // rows per group
        constexpr int NG = MI / GM;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int fo = kk ? fr_off1 : fr_off0;
            f32x4 fb[NI];
            f32x4 fa[2][GM];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const f32x4*>(stage + b_base + ni * 2048 + fo);
#pragma unroll
            for (int i = 0; i < GM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(stage + a_base + i * 2048 + fo);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < GM; ++i)
                        fa[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(stage + a_base + ((g + 1) * GM + i) * 2048 + fo);
                }
#pragma unroll
                for (int i = 0; i < GM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[g * GM + i][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[ni][j], fa[g & 1][i][j], acc[g * GM + i][ni], 0, 0, 0);
            }
            // pin the issue order the source spells out (hipcc otherwise sinks each read next to its first use to save
            // registers): operand B and the first row group, then one read of the next group per row of MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, NI + GM, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < GM; ++i) {
                    if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI, 0);
                }
        }
    }

    gemm_epilogue_f32<MI, NI, WM, WN>(p, acc, smem_x32, m0, n0, z, lid, wave, lane, HoWo);
}

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg_2x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm2x_kernel<MI, NI, WM, WN>;
    constexpr size_t lds = 2 * (size_t)(16 * MI * WM + 16 * NI * WN) * 128;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm2x(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesX) return hipErrorInvalidValue;
    if ((p.Cin % 32) || p.CS != 32 || !p.zero_page || p.out_mode != 0) return hipErrorInvalidValue;
    if (p.geglu && (cfg == 0 || cfg == 3 || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;  // needs an even NI
    const int bm = gemm_tile_info_x(cfg).bm, bn = gemm_tile_info_x(cfg).bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const int tiles = MT * NT;
    const dim3 grid = gemm_grid(p, tiles);
    switch (cfg) {
        case 0: return launch_cfg_2x<8, 5, 2, 4>(p, grid, stream);
        case 1: return launch_cfg_2x<8, 4, 2, 4>(p, grid, stream);
        case 2: return launch_cfg_2x<4, 4, 4, 2>(p, grid, stream);
        case 3: return launch_cfg_2x<4, 5, 2, 4>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
