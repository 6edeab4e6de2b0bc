// k_gemm_bf16p.hip -- "ping-pong" large-tile bf16 implicit-GEMM conv / linear (precision = 1; bf16 tile_cfg 104 + x).
//
// k_gemm_bf16x.hip keeps both waves of a SIMD in the same phase: each interleaves its fragment reads, its share of the LDS-DMA and its
// matrix instructions, and the k loop holds the matrix pipe ~59 % busy (a 256 x 320 tile at K = 2880: 98 us of k loop against 57.6 us of
// matrix time at the 2.0 GHz the chip holds).  Here the eight waves are two groups of four -- waves 0-3 and 4-7, i.e. the two waves that
// share each SIMD are in different groups -- running the same two-phase program half a period apart:
//     LOAD(s):     read the wave tile's fragments of K = 32 slab s from LDS (MI + NI ds_read_b128), issue this wave's share of the
//                  LDS-DMA of slab s + 3, wait for the reads and for the wave's own share of slab s + 1 (counted vmcnt), s_barrier
//     COMPUTE(s):  MI x NI back-to-back v_mfma_f32_16x16x32_bf16 at s_setprio 1, s_barrier
// While one wave of a SIMD computes, its partner loads, so the matrix pipe sees one uninterrupted instruction stream and the LDS / DMA
// work has a whole compute phase to complete in.  A wave needs its fragments single-buffered only (MI + NI register quads).
// LDS is a ring of four K = 32 slabs of (BM + BN) rows x 64 B (= 144 KB for 256 x 320); a slab is staged as 16-row x 64-B pieces (one
// global_load_lds_dwordx4 = 1 KiB per wave), lane -> row lane >> 2, slot lane & 3 <- the row's 16-byte chunk (lane & 3) ^ f(row),
// f(r) = (-(r >> 2)) & 3 -- the piece layout and swizzle of k_gemm3p.hip, conflict-free for the b128 fragment reads; a piece IS a
// 16 x 32 fragment.  Slab s = half (s & 1) of the 64-channel k tile s >> 1 of k_gemm_bf16x.hip, so both kernels read the same packed
// weights and NHWC activations.
// Intervals (one s_barrier each, all eight waves): group 0 runs LOAD(s) in interval 2 s and COMPUTE(s) in 2 s + 1, group 1 LOAD(s) in
// 2 s + 1 and COMPUTE(s) in 2 s + 2.  Hazards: slab s is read in intervals 2 s and 2 s + 1, and those reads are waited for (lgkmcnt(0))
// BEFORE the barrier that ends the interval; the DMA into its slot (slab s + 4) is issued in intervals 2 s + 2 / 2 s + 3.  A wave waits
// for its OWN share of slab s + 1 at the end of LOAD(s) (vmcnt = two slabs' worth of its DMA instructions may stay in flight; in the
// tail, where fewer were issued, vmcnt(0)), i.e. before the barriers that end intervals 2 s and 2 s + 1, and slab s + 1 is first read
// in interval 2 s + 2.
// Epilogue, tile map, split-K: shared with k_gemm_bf16x.hip (k_gemm_bf16_epi.hpp).
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

static const GemmTileInfo kTilesPP[kNumGemmTilesPP] = {{256, 320, "256x320pp"}, {256, 256, "256x256pp"}, {256, 320, "256x320pc"}, {256, 256, "256x256pc"}};
const GemmTileInfo& gemm_tile_info_pp(int cfg) { return kTilesPP[cfg]; }

template <int N>
__device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pp_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The k loop of one wave.  GROUP 0 = waves 0-3 (lead), GROUP 1 = waves 4-7 (half a period behind).
// PROBE (diagnostic instantiation behind option gemm_probe): every wave sums the shader-clock cycles of its segments over the k loop --
// fragment reads issued / DMA issued / wait for the reads / wait for the DMA / barrier after LOAD / matrix phase / barrier after COMPUTE / the
// whole loop -- into probe[(block * 8 + wave) * 8 + segment] (blocks 0..1023 of slice 0).
template <int MI, int NI, int WN, int GROUP, bool PROBE, bool DMAC>
__device__ __forceinline__ void pp_kloop(const ConvGemm& p, f32x4 (&acc)[MI][NI], unsigned char* smem, const int m0, const int n0,
                                         const int kt_begin, const int n_t, const int wave, const int lane, const int HoWo) {
    constexpr int BM = 16 * MI * 2;
    constexpr int BN = 16 * NI * WN;
    constexpr int PA = BM / 16;              // activation pieces of a slab
    constexpr int PB = BN / 16;              // weight pieces
    constexpr int SLAB = (PA + PB) * 1024;   // bytes
    constexpr int NAJ = PA / 8;              // activation pieces per wave
    constexpr int NBJ = (PB % 8 == 0) ? PB / 8 : (GROUP == 0 ? PB / 8 + 1 : PB / 8);   // weight pieces per wave (PB = 20: waves 0-3 take three)
    constexpr int NP = NAJ + NBJ;            // DMA instructions per wave per slab
    static_assert(PA % 8 == 0 && (PB % 8 == 0 || PB % 8 == 4), "pieces per wave");
    constexpr int WNC = 16 * NI;

    const int wm = wave / WN;                // == GROUP
    const int wn = wave - wm * WN;
    const int S = 2 * n_t;                   // slabs of this k slice
    const int Hin = p.Hs << p.ups, Win = p.Ws << p.ups;
    const unsigned pix_bytes = (unsigned)p.a_ld * 2u;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);
    const bool geglu = p.geglu != 0;

    // pieces: lane -> row lane >> 2 of the 16-row group, LDS slot lane & 3 <- the row's 16-byte chunk (lane & 3) ^ f(row)
    const bool wide = (p.variant & 8) != 0;     // DIAGNOSTIC (wrong results): lanes address 8 rows x 128 B instead of 16 rows x 64 B
    const int r16 = wide ? (lane >> 3) : (lane >> 2);
    const int ch = wide ? (lane & 7) : ((lane & 3) ^ ((-(r16 >> 2)) & 3));
    int a_iy0[NAJ], a_ix0[NAJ];
    unsigned a_off[NAJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int m = m0 + (wave + 8 * j) * 16 + r16;
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
    unsigned b_off[NBJ];
    bool b_ok[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int f = wave + 8 * j;          // fragment group (16 rows) of the weight tile
        int n = n0 + f * 16 + r16;
        long long wrow = n;
        if (geglu) {
            const int fw = f / NI, ni = f - fw * NI;
            n = n0 + fw * (WNC / 2) + (ni >> 1) * 16 + r16;
            wrow = (long long)n + ((ni & 1) ? p.N : 0);
        }
        b_ok[j] = n < p.N;
        b_off[j] = b_ok[j] ? (unsigned)wrow * ((unsigned)p.b_ld * 2u) + ch * 16 : 0u;
    }

    const int T = p.KH * p.KW;
    int cs = kt_begin / T;
    const int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt = kt_begin, hh = 0, s_issue = 0;

    // DMA instruction J (of NP) of this wave's share of slab s_issue; the last one advances (cs, ky, kx, kt, hh, s_issue)
    auto piece = [&](auto JC) {
        constexpr int J = decltype(JC)::value;
        unsigned char* dst = smem + (s_issue & 3) * SLAB;
        if constexpr (J < NAJ) {
            const int iy = a_iy0[J] + ky;
            const int ix = a_ix0[J] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const unsigned off = a_off[J] + (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups)) * pix_bytes + (unsigned)cs * 128u + (unsigned)hh * 64u;
            const char* src = (ok ? Abase : zero) + (ok ? off : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(dst + (wave + 8 * J) * 1024), 16, 0, 0);
        } else {
            constexpr int j = J - NAJ;
            const unsigned wk = (unsigned)kt * 128u + (unsigned)hh * 64u;
            const char* src = (b_ok[j] ? Bbase : zero) + (b_ok[j] ? b_off[j] + wk : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(dst + (PA + wave + 8 * j) * 1024), 16, 0, 0);
        }
        if constexpr (J == NP - 1) {
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
        }
    };
    auto issue = [&]() {      // this wave's whole share of slab s_issue
        piece(std::integral_constant<int, 0>{});
        piece(std::integral_constant<int, 1>{});
        if constexpr (NP > 2) piece(std::integral_constant<int, 2>{});
        if constexpr (NP > 3) piece(std::integral_constant<int, 3>{});
        if constexpr (NP > 4) piece(std::integral_constant<int, 4>{});
        static_assert(NP <= 5, "pieces per wave");
    };

    // fragment reads: row c of a piece, slot g ^ f(c)
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr = c15 * 64 + ((g4 ^ ((-(c15 >> 2)) & 3)) << 4);
    const int a_fr = wm * MI * 1024 + fr;
    const int b_fr = (PA + wn * NI) * 1024 + fr;

    // prologue: slabs 0, 1, 2; own share of slab 0 landed; barrier 0
    issue();
    if (S > 1) issue();
    if (S > 2) issue();
    if (S > 2) pp_wait_vm<2 * NP>(); else pp_wait_vm<0>();
    pp_barrier();
    if constexpr (GROUP == 1) pp_barrier();      // interval 0: group 0 reads slab 0

    u32x4 fa[MI], fb[NI];
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0, tl = 0;
    auto stamp = [&](int seg) {
        if constexpr (PROBE) { t1 = __builtin_amdgcn_s_memtime(); pt[seg] += t1 - t0; t0 = t1; }
    };
    if constexpr (PROBE) { t0 = __builtin_amdgcn_s_memtime(); tl = t0; }
    for (int s = 0; s < S; ++s) {
        // ---- LOAD(s)
        const unsigned char* sl = smem + (s & 3) * SLAB;
        if (!(p.variant & 4) || s == 0) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(sl + b_fr + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const u32x4*>(sl + a_fr + mi * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(0);
        const bool more = s + 3 < S && !(p.variant & 1);
        if constexpr (!DMAC) {
            if (more) issue();
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);
        pp_wait_lgkm0();
        stamp(2);
        if constexpr (DMAC) {     // issued so far: slabs .. s + 2; slab s + 2 may stay in flight
            if (s + 2 < S && !(p.variant & 1)) pp_wait_vm<NP>(); else pp_wait_vm<0>();
        } else {
            if (more) pp_wait_vm<2 * NP>(); else pp_wait_vm<0>();
        }
        stamp(3);
        pp_barrier();
        stamp(4);
        // ---- COMPUTE(s)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (DMAC) {
            // the DMA of slab s + 3 rides in the matrix phase: one piece behind each of the first NP fragment rows
            auto row = [&](auto MC) {
                constexpr int mi = decltype(MC)::value;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]), acc[mi][ni], 0, 0, 0);
                if constexpr (mi < NP) { if (more) piece(MC); }
                __builtin_amdgcn_sched_barrier(0);
            };
            static_assert(MI == 8, "fragment rows");
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
        } else if (!(p.variant & 2)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[mi]), acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(5);
        if (GROUP == 0 || s + 1 < S) pp_barrier();     // (group 1's last COMPUTE runs into the epilogue's barrier)
        stamp(6);
    }
    if constexpr (PROBE) {
        pt[7] = __builtin_amdgcn_s_memtime() - tl;
        if (p.probe && blockIdx.x < 1024 && blockIdx.z == 0 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.probe[((unsigned long long)blockIdx.x * 8 + wave) * 8 + i] = pt[i];
        }
    }
}

template <int MI, int NI, int WN, bool PROBE = false, bool DMAC = false>
__global__ __launch_bounds__(512) void conv_gemm_bf16p_kernel(const ConvGemm p) {
    constexpr int WM = 2;
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    static_assert(WN == 4, "waves 0-3 / 4-7 = the two groups");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pp[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const int n_t = kt_end - kt_begin;
    const int HoWo = p.Ho * p.Wo;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave < 4) pp_kloop<MI, NI, WN, 0, PROBE, DMAC>(p, acc, smem_pp, m0, n0, kt_begin, n_t, wave, lane, HoWo);
    else pp_kloop<MI, NI, WN, 1, PROBE, DMAC>(p, acc, smem_pp, m0, n0, kt_begin, n_t, wave, lane, HoWo);

    gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_pp, m0, n0, z, wave, lane, HoWo);
}

template <int MI, int NI, int WN, bool PROBE = false, bool DMAC = false>
static hipError_t launch_cfg_bf16p(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16p_kernel<MI, NI, WN, PROBE, DMAC>;
    constexpr size_t lds = 4 * (size_t)(2 * MI + NI * WN) * 1024;
    static_assert(lds <= 160 * 1024, "the slab ring must fit the CU's LDS");
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_bf16p(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesPP) return hipErrorInvalidValue;
    if ((p.Cin % 64) || !p.zero_page) return hipErrorInvalidValue;
    if (p.geglu && (cfg == 0 || cfg == 2 || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;  // needs an even NI
    const int bm = kTilesPP[cfg].bm, bn = kTilesPP[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const dim3 grid = gemm_grid(p, MT * NT);
    if (p.probe) return cfg == 0 ? launch_cfg_bf16p<8, 5, 4, true>(p, grid, stream) : cfg == 2 ? launch_cfg_bf16p<8, 5, 4, true, true>(p, grid, stream) : hipErrorInvalidValue;
    switch (cfg) {
        case 0: return launch_cfg_bf16p<8, 5, 4>(p, grid, stream);
        case 1: return launch_cfg_bf16p<8, 4, 4>(p, grid, stream);
        case 2: return launch_cfg_bf16p<8, 5, 4, false, true>(p, grid, stream);     // the same tiles with the DMA issued in the matrix phase
        case 3: return launch_cfg_bf16p<8, 4, 4, false, true>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
