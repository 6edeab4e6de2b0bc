// k_gemm_bf16s.hip -- 128-row, 4-wave bf16 implicit-GEMM conv / linear tiles with TWO workgroups per CU (precision = 1; bf16 tile_cfg 104 + x).
//
// Why: the transformer blocks' Linear layers are short-K GEMMs over many rows (K = 320 ... 1280, M = 8 192 ... 131 072 at batch 16): 45 % of
// the bf16 GEMM time at 0.52-1.0 PFLOP/s (profiles/r04g_*).  On the 256-row tiles of k_gemm_bf16x.hip a K = 320 tile is five k tiles of
// matrix work between a DMA prologue and a 164 KB store that nothing overlaps (one workgroup per CU): with the k loop EMPTY such a launch
// still takes 72 % of its time (profiles/r04c_*), and the 64^2-level layers are HBM-bound besides (N = 320: 168 MB per 26.8 GFLOP).
// Here a workgroup is 4 waves -- one per SIMD -- on a 128 x 320 (or 128 x 256) tile, each wave the same 128 x 80 wave tile as before
// (160 accumulators), and its LDS is two K = 32 slabs of (128 + BN) rows x 64 B = 56 KB, so two workgroups share a CU: one's prologue and
// store run beside the other's k loop, and twice as many memory requests are in flight per CU.  The price is 1.56x the staged bytes per
// flop of the 256-row tile, which a memory-bound layer does not notice and a long-K convolution does (those keep the 256-row tiles).
// A slab is staged as 16-row x 64-B LDS-DMA pieces (lane -> row lane >> 2, slot lane & 3 <- chunk (lane & 3) ^ f(row), f(r) = (-(r >> 2)) & 3:
// the conflict-free piece layout of k_gemm3p.hip; a piece is a 16 x 32 fragment); slab s = half (s & 1) of the 64-channel k tile s >> 1, so
// the packed weights and NHWC activations are those of the other bf16 kernels.  Loop per slab: wait for the own share of slab s, s_barrier
// (slab s complete, everybody is done reading slab s - 1), issue the DMA of slab s + 1 into the other buffer, MI + NI fragment reads,
// MI x NI matrix instructions.  Epilogue, tile map and split-K are shared (k_gemm_bf16_epi.hpp).
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_bf16_epi.hpp"
#include <type_traits>

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

static const GemmTileInfo kTilesS2[kNumGemmTilesS2] = {{128, 320, "128x320s"}, {128, 256, "128x256s"}};
const GemmTileInfo& gemm_tile_info_s2(int cfg) { return kTilesS2[cfg]; }

template <int NI>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16s_kernel(const ConvGemm p) {
    constexpr int MI = 8, WM = 1, WN = 4;
    constexpr int BM = 16 * MI * WM;         // 128
    constexpr int BN = 16 * NI * WN;         // 320 / 256
    constexpr int PA = BM / 16;              // activation pieces of a slab (8)
    constexpr int PB = BN / 16;              // weight pieces (20 / 16)
    constexpr int SLAB = (PA + PB) * 1024;   // bytes
    constexpr int NAJ = PA / 4;              // per wave: 2
    constexpr int NBJ = PB / 4;              // 5 / 4
    constexpr int WNC = 16 * NI;
    static_assert(2 * SLAB <= 80 * 1024, "two workgroups per CU");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = wn (WM = 1)

    const bool geglu = p.geglu != 0;
    const int BNO = geglu ? BN / 2 : BN;   // output columns per tile
    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BNO - 1) / BNO;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int m0 = gw.tm * BM;
    const int n0 = gw.tn * BNO;
    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int S = 2 * (kt_end - kt_begin);   // slabs of this k slice
    const int HoWo = p.Ho * p.Wo;

    const int Hin = p.Hs << p.ups, Win = p.Ws << p.ups;
    const unsigned pix_bytes = (unsigned)p.a_ld * 2u;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // pieces: lane -> row lane >> 2 of the 16-row group, LDS slot lane & 3 <- the row's 16-byte chunk (lane & 3) ^ f(row)
    const int r16 = lane >> 2;
    const int ch = (lane & 3) ^ ((-(r16 >> 2)) & 3);
    int a_iy0[NAJ], a_ix0[NAJ];
    unsigned a_off[NAJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int m = m0 + (wave + 4 * j) * 16 + r16;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_off[j] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * pix_bytes + ch * 16;
        a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
        a_ix0[j] = ox * p.stride - p.pad;
    }
    unsigned b_off[NBJ];                     // ~0u: row beyond N -> zero page
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int f = wave + 4 * j;          // fragment group (16 rows) of the weight tile
        int n = n0 + f * 16 + r16;
        long long wrow = n;
        if (geglu) {
            const int fw = f / NI, ni = f - fw * NI;
            n = n0 + fw * (WNC / 2) + (ni >> 1) * 16 + r16;
            wrow = (long long)n + ((ni & 1) ? p.N : 0);
        }
        b_off[j] = n < p.N ? (unsigned)wrow * ((unsigned)p.b_ld * 2u) + ch * 16 : ~0u;
    }

    const int T = p.KH * p.KW;
    int cs = kt_begin / T;
    const int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt = kt_begin, hh = 0, s_issue = 0;

    auto issue = [&]() {      // this wave's share of slab s_issue -> buffer s_issue & 1
        unsigned char* dst = smem_s2 + (s_issue & 1) * SLAB;
        const unsigned koff = (unsigned)cs * 128u + (unsigned)hh * 64u;
#pragma unroll
        for (int j = 0; j < NAJ; ++j) {
            const int iy = a_iy0[j] + ky;
            const int ix = a_ix0[j] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const unsigned off = a_off[j] + (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups)) * pix_bytes + koff;
            const char* src = (ok ? Abase : zero) + (ok ? off : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(dst + (wave + 4 * j) * 1024), 16, 0, 0);
        }
        const unsigned wk = (unsigned)kt * 128u + (unsigned)hh * 64u;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const bool ok = b_off[j] != ~0u;
            const char* src = (ok ? Bbase : zero) + (ok ? b_off[j] + wk : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(dst + (PA + wave + 4 * j) * 1024), 16, 0, 0);
        }
        if (hh) {
            const bool wrap_x = (kx + 1 == p.KW);
            const bool wrap_y = wrap_x && (ky + 1 == p.KH);
            kx = wrap_x ? 0 : kx + 1;
            ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
            cs = wrap_y ? cs + 1 : cs;
            ++kt;
        }
        hh ^= 1;
        ++s_issue;
    };

    // fragment reads: row c of a piece, slot g ^ f(c)
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr = c15 * 64 + ((g4 ^ ((-(c15 >> 2)) & 3)) << 4);
    const int b_fr = (PA + wave * NI) * 1024 + fr;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue();                                  // slab 0
    u32x4 fa[MI], fb[NI];
    for (int s = 0; s < S; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of slab s (the only DMA it has in flight)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();         // slab s is complete; every wave has read slab s - 1 (lgkmcnt(0) below) out of the other buffer
        asm volatile("" ::: "memory");
        if (s + 1 < S) issue();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sl = smem_s2 + (s & 1) * SLAB;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(sl + b_fr + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(sl + fr + mi * 1024);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]), acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_s2, m0, n0, z, wave, lane, HoWo);
}

template <int NI>
static hipError_t launch_cfg_bf16s(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16s_kernel<NI>;
    constexpr size_t lds = 2 * (size_t)(8 + NI * 4) * 1024;
    // (the epilogue transposes through one 16 x (16 NI + 4) fp32 scratch per wave in the same memory: 4 x 5.4 KB)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_bf16s(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesS2) return hipErrorInvalidValue;
    if ((p.Cin % 64) || !p.zero_page) return hipErrorInvalidValue;
    if (p.geglu && (cfg == 0 || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;  // needs an even NI
    const int bm = kTilesS2[cfg].bm, bn = kTilesS2[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const dim3 grid = gemm_grid(p, MT * NT);
    switch (cfg) {
        case 0: return launch_cfg_bf16s<5>(p, grid, stream);
        case 1: return launch_cfg_bf16s<4>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
