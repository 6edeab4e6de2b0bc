// k_gemm3p.hip -- fp32 implicit-GEMM conv / linear on the bf16 matrix pipe with BOTH operands as three bf16 planes ("plane" kernel,
// precision = 0; tile_cfg 300 + x).
//
// The arithmetic is k_gemm3x.hip's: an fp32 number is exactly h + m + l with h, m, l bf16 (k_split3.hpp), a bf16 x bf16 product is
// exact in fp32, and  a w  is accumulated in fp32 as the six partial products  wl ah, wh al, wm am, wm ah, wh am, wh ah  (smallest
// first; the three dropped ones are below 2^-23 |a w|: tests/test_split_oracle_cpu.py).  What differs is WHERE the activations are
// split.  k_gemm3x.hip stages them as fp32 and splits every fragment in registers inside the k loop: 176 of the 430 instructions a wave
// issues per k tile around its 120 matrix instructions, redone by every wave that shares the rows, for every tap of a 3x3 convolution
// and every tile column (PMC round 2: matrix pipe 53-63 % busy, 17 % on the K = 320 linears).  Here the PRODUCER of an activation
// (GroupNorm / LayerNorm apply, the GEGLU gate, the attention kernel, a GEMM epilogue, or split3_rows_kernel below for tensors that arrive
// as fp32) writes the three planes once -- [pixel][C / 32][plane h, m, l][32] bf16, the layout of the weight planes -- and this kernel
// moves them HBM -> LDS by LDS-DMA exactly like the weights: per k tile a wave issues 3 + 3 NI plane reads per fragment row pair, its share
// of the DMA, and 6 MI NI matrix instructions.  No VALU work in the k loop.
//
// LDS stage = (BM / 16) x 3 activation pieces + (BN / 16) x 3 weight pieces of 1 KiB; a piece = one plane of one 16-row fragment group =
// 16 rows x 64 B, lane -> row lane >> 2, slot lane & 3 receives the plane's 16-byte chunk (lane & 3) ^ f(row), f(r) = (-(r >> 2)) & 3
// (the swizzle of k_gemm3x.hip's weight pieces: conflict-free ds_read_b128 of 64-byte rows).  A wave owns ALL THREE planes of its
// activation fragment groups, so the gather arithmetic (tap, padding, nearest-2x upsample, zero page) is done once per group.
// Everything else -- XCD-aware tile map, deterministic split-K slabs, the epilogue (k_gemm_epi.hpp), the k order (channel slice outer,
// taps inner, chunk g of a plane row = elements 4g..4g+3, 16+4g..16+4g+3) -- is shared with k_gemm3x.hip, so both read the same weight planes.
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_epi.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

static const GemmTileInfo kTilesP[kNumGemmTilesP] = {
    {256, 160, "256x160p"}, {256, 128, "256x128p"}, {128, 256, "128x256p"}, {128, 160, "128x160p"}, {128, 128, "128x128p"},
    {64, 64, "64x64p"}, {64, 128, "64x128p"}, {64, 320, "64x320p"}, {128, 64, "128x64p"}};
const GemmTileInfo& gemm_tile_info_p(int cfg) { return kTilesP[cfg]; }

// Per-wave state of the k loop; every array is indexed with compile-time constants (member templates) and lives in registers.  The
// issue order of a k tile is spelled out and fenced with sched_barrier(0) as in k_gemm3x.hip: with LDS-DMA in flight every wait hipcc
// inserts is lgkmcnt(0), so a plane read is issued NI matrix instructions or more ahead of its first use and never right in front of it.
template <int MI, int NI, int NAG, int NBW, int A_BYTES, int NWV>
struct P3Wave {
    static constexpr int NA = 3 * NAG;          // activation pieces per wave per k tile (NAG fragment groups x 3 planes)
    static constexpr int NP = NA + NBW;         // DMA instructions per wave per k tile
    static constexpr int NMF = 6 * NI;          // matrix instructions of one fragment row
    static constexpr int NAF = MI > 2 ? 3 : MI; // activation fragment buffers (row r uses buffer r % NAF)
    static_assert(MI == 2 || MI == 4, "fragment rows per wave");
    static_assert(NI >= 2, "plane reads of a fragment take three instruction slots (behind())");

    f32x4 acc[MI][NI];
    u32x4 wf[3][NI];            // weight planes h, m, l of the wave's NI column fragments
    u32x4 af[NAF][3];           // activation planes h, m, l of a fragment row
    int a_iy0[NAG], a_ix0[NAG];
    unsigned a_off[NAG];        // byte offset of the sample + this lane's chunk (operands are < 4 GiB: launch side checks)
    unsigned w_off[NBW];
    unsigned w_kstep;           // bytes between a weight piece's k tiles: 192 (row-major planes) or 3072 (16-row groups, ConvGemm::b3_grouped)
    const char *Abase, *Wbase, *zero, *a_src;
    unsigned pix_bytes;
    int Hin, Win, ups, Ws, KH, KW, wave;
    int cs, ky, kx, kt_next, kt_end;
    const unsigned char* a_tile;    // stage + this wave's activation pieces + lane offset
    const unsigned char* w_tile;    // stage + A_BYTES + this wave's weight pieces + lane offset
    unsigned char* next_stage;      // where the DMA of k tile kt_next goes

    // DMA instruction J of k tile kt_next -> next_stage.  Straight-line code (selects, no branches); weight pieces past the tile's
    // last one re-fetch the last row, and after the last k tile the same tile is fetched once more into the stage nobody reads.
    template <int J>
    __device__ __forceinline__ void piece() {
        if constexpr (J < NA) {
            constexpr int jg = J / 3, pl = J - 3 * jg;
            if constexpr (pl == 0) {
                const int iy = a_iy0[jg] + ky;
                const int ix = a_ix0[jg] + kx;
                const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const unsigned off = a_off[jg] + (unsigned)((iy >> ups) * Ws + (ix >> ups)) * pix_bytes + (unsigned)cs * 192u;
                a_src = (ok ? Abase : zero) + (ok ? off : 0u);
            }
            __builtin_amdgcn_global_load_lds((global_cvoid*)(a_src + pl * 64), (lds_void*)(next_stage + ((wave + NWV * jg) * 3 + pl) * 1024), 16, 0, 0);
        } else {
            constexpr int j = J - NA;
            const char* src = Wbase + (w_off[j] + (unsigned)kt_next * w_kstep);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(next_stage + A_BYTES + (wave + NWV * j) * 1024), 16, 0, 0);
        }
        if constexpr (J == NP - 1) {
            const bool adv = kt_next + 1 < kt_end;
            const bool wrap_x = (kx + 1 == KW);
            const bool wrap_y = wrap_x && (ky + 1 == KH);
            const int kx1 = wrap_x ? 0 : kx + 1;
            const int ky1 = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
            const int cs1 = wrap_y ? cs + 1 : cs;
            kx = adv ? kx1 : kx;
            ky = adv ? ky1 : ky;
            cs = adv ? cs1 : cs;
            kt_next = adv ? kt_next + 1 : kt_next;
        }
    }
    template <int J0, int J1>
    __device__ __forceinline__ void pieces() {
        if constexpr (J0 < J1) { piece<J0>(); pieces<J0 + 1, J1>(); }
    }

    template <int F, int PL>
    __device__ __forceinline__ void read_a() { af[F % NAF][PL] = *reinterpret_cast<const u32x4*>(a_tile + (F * 3 + PL) * 1024); }
    template <int PL, int N>
    __device__ __forceinline__ void read_w() { wf[PL][N] = *reinterpret_cast<const u32x4*>(w_tile + (N * 3 + PL) * 1024); }

    // What is issued behind matrix instruction K of fragment row MIDX.
    //   row 0 reads the tile's operands just in time: before it the h plane of fragment 0 and the l weight planes (product 0 = wl ah);
    //   behind the NI instructions of product 0 the h weight planes + fragment 0's l (product 1 = wh al), behind those of product 1 the m
    //   weight planes + fragment 0's m (product 2 = wm am), behind the first instructions of product 2 fragment 1, then fragment 2;
    //   row r >= 1 reads fragment r + 2 into the buffer row r - 1 has just released;
    //   the next k tile's DMA instructions are spread over the slots of row 0's products 3..5 and of row 1.
    static constexpr int DMA_SLOTS = 3 * NI + NMF;
    template <int MIDX, int K>
    __device__ __forceinline__ void behind() {
        constexpr int pr = K / NI, ni = K % NI;
        if constexpr (MIDX == 0) {
            if constexpr (pr == 0) { read_w<0, ni>(); if constexpr (ni == NI - 1) read_a<0, 2>(); }
            if constexpr (pr == 1) { read_w<1, ni>(); if constexpr (ni == NI - 1) read_a<0, 1>(); }
            // the three planes of fragment 1 behind instructions 2 NI .. 2 NI + 2, those of fragment 2 behind the next three (NI >= 2)
            if constexpr (K >= 2 * NI && K < 2 * NI + 3) read_a<1, K - 2 * NI>();
            if constexpr (MI > 2 && K >= 2 * NI + 3 && K < 2 * NI + 6) read_a<2, K - 2 * NI - 3>();
        } else if constexpr (MIDX + 2 < MI) {
            if constexpr (K < 3) read_a<MIDX + 2, K>();
        }
        if constexpr (MIDX == 0 && pr >= 3) {
            constexpr int slot = K - 3 * NI;
            pieces<slot * NP / DMA_SLOTS, (slot + 1) * NP / DMA_SLOTS>();
        } else if constexpr (MIDX == 1) {
            constexpr int slot = 3 * NI + K;
            pieces<slot * NP / DMA_SLOTS, (slot + 1) * NP / DMA_SLOTS>();
        }
    }
    template <int MIDX, int K>
    __device__ __forceinline__ void mfmas() {
        if constexpr (K < NMF) {
            constexpr int WP[6] = {2, 0, 1, 1, 0, 0};      // weight plane of product K / NI
            constexpr int AP[6] = {0, 2, 1, 0, 1, 0};      // activation plane
            constexpr int pr = K / NI, ni = K % NI;
            acc[MIDX][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[WP[pr]][ni]), __builtin_bit_cast(bf16x8, af[MIDX % NAF][AP[pr]]),
                                                                    acc[MIDX][ni], 0, 0, 0);
            behind<MIDX, K>();
            __builtin_amdgcn_sched_barrier(0);
            mfmas<MIDX, K + 1>();
        }
    }
    template <int MIDX>
    __device__ __forceinline__ void rows() {
        if constexpr (MIDX < MI) { mfmas<MIDX, 0>(); rows<MIDX + 1>(); }
    }
    template <int N>
    __device__ __forceinline__ void head_w() {
        if constexpr (N < NI) { read_w<2, N>(); head_w<N + 1>(); }
    }
    __device__ __forceinline__ void tile() {
        read_a<0, 0>();
        head_w<0>();
        __builtin_amdgcn_sched_barrier(0);
        rows<0>();
    }
};

// WM x WN = 8 waves: the large tiles (one workgroup per CU).  WM x WN = 4 waves: 64-row / 64-column tiles whose stages are 24-36 KB, so that
// two or three workgroups share a CU and one's DMA prologue / epilogue hides behind the others' matrix work -- for the K = 320 ... 1280
// linears of the transformer blocks, which as 128-row tiles needed split-K slabs + a reduce launch to fill the chip (round 2: 5 200
// launches per image at < 20 % matrix-pipe use).
// PROBE (diagnostic instantiations behind option gemm_probe, tools/probes/gemm_phase_probe.py): every workgroup stores s_memrealtime stamps
// (100 MHz) of kernel entry / first k tile landed / k loop done / epilogue stores acknowledged, and every wave the shader-clock cycles it spent
// in the k loop and, of those, waiting at the per-tile barrier: probe[24 * block + {0..3, 4 + 2 wave, 5 + 2 wave}].
template <int MI, int NI, int WM, int WN, int NSTG, bool PROBE = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm3p_kernel(const ConvGemm p) {
    static_assert(NSTG == 2 || NSTG == 3, "LDS stages");
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int NWV = WM * WN;
    static_assert(NWV == 8 || NWV == 4, "8 or 4 waves per workgroup");
    static_assert(BM % (16 * NWV) == 0, "every wave owns whole 16-row fragment groups of the activation tile");
    constexpr int NAG = BM / (16 * NWV);      // activation fragment groups (16 rows x 3 planes) per wave per k tile
    constexpr int PW = (BN / 16) * 3;         // weight pieces per k tile
    constexpr int NBW = (PW + NWV - 1) / NWV; // ... per wave
    constexpr int A_BYTES = (BM / 16) * 3 * 1024;
    constexpr int STAGE = A_BYTES + NBW * NWV * 1024;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p3[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    constexpr int WNC = 16 * NI;
    const bool geglu = p.geglu != 0;
    const int BNO = geglu ? BN / 2 : BN;
    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BNO - 1) / BNO;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int lid = gw.lid;
    const int m0 = gw.tm * BM;
    const int n0 = gw.tn * BNO;
    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;
    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;

    P3Wave<MI, NI, NAG, NBW, A_BYTES, NWV> w;
    w.Hin = p.Hs << p.ups;
    w.Win = p.Ws << p.ups;
    w.ups = p.ups;
    w.Ws = p.Ws;
    w.KH = p.KH;
    w.KW = p.KW;
    w.wave = wave;
    w.pix_bytes = (unsigned)p.a3_ld;
    w.Abase = reinterpret_cast<const char*>(p.A3);
    w.Wbase = reinterpret_cast<const char*>(p.Bt3);
    w.zero = reinterpret_cast<const char*>(p.zero_page);
    w.a_src = w.zero;

    // pieces: lane -> row lane >> 2 of the 16-row group, LDS slot lane & 3 <- the plane row's 16-byte chunk (lane & 3) ^ f(row)
    const int r16 = lane >> 2;
    const int ch = (lane & 3) ^ ((-(r16 >> 2)) & 3);
#pragma unroll
    for (int j = 0; j < NAG; ++j) {
        const int m = m0 + (wave + NWV * j) * 16 + r16;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        w.a_off[j] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * w.pix_bytes + ch * 16;
        w.a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
        w.a_ix0[j] = ox * p.stride - p.pad;
    }
    const unsigned w_row_bytes = (unsigned)p.kt_total * 192u;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int q = wave + NWV * j;
        const int f = q / 3, pl = q - 3 * f;
        int n = n0 + f * 16 + r16;
        long long wrow = n;
        if (p.geglu == 2) {        // wave columns [0, WN / 2) hold the value rows of the tile's outputs, [WN / 2, WN) their gate rows (k_gemm_epi.hpp)
            const int fw = f / NI, ni = f - fw * NI;
            const int vg = fw / (WN / 2 > 0 ? WN / 2 : 1), col = fw - vg * (WN / 2);
            n = n0 + col * WNC + ni * 16 + r16;
            wrow = (long long)n + (vg ? p.N : 0);
        } else if (geglu) {
            const int fw = f / NI, ni = f - fw * NI;
            n = n0 + fw * (WNC / 2) + (ni >> 1) * 16 + r16;
            wrow = (long long)n + ((ni & 1) ? p.N : 0);
        }
        // rows past N (ragged last tile) and the pieces past PW fetch the last valid row: real memory, never stored
        if (n >= p.N) wrow -= (n - (p.N - 1));
        w.w_off[j] = p.b3_grouped ? (unsigned)(wrow >> 4) * (w_row_bytes * 16u) + pl * 1024 + (unsigned)(wrow & 15) * 64 + ch * 16
                                  : (unsigned)wrow * w_row_bytes + pl * 64 + ch * 16;
    }
    w.w_kstep = p.b3_grouped ? 3072u : 192u;

    w.cs = kt_begin / T;
    const int tap0 = kt_begin - w.cs * T;
    w.ky = tap0 / p.KW;
    w.kx = tap0 - w.ky * p.KW;
    w.kt_next = kt_begin;
    w.kt_end = kt_end;

    // fragment reads: row c of a piece, slot g ^ f(c)
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr = c15 * 64 + ((g4 ^ ((-(c15 >> 2)) & 3)) << 4);
    const int a_fr = wm * MI * 3 * 1024 + fr;
    const int w_fr = A_BYTES + wn * NI * 3 * 1024 + fr;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) w.acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    unsigned long long pt0 = 0, pt1 = 0, pc0 = 0, pwait = 0;
    if constexpr (PROBE) pt0 = __builtin_amdgcn_s_memrealtime();
    w.next_stage = smem_p3;
    w.template pieces<0, NAG * 3 + NBW>();      // k tile 0
    if constexpr (PROBE) {                      // the wait for k tile 0, taken out of the loop's first barrier
        __syncthreads();
        pt1 = __builtin_amdgcn_s_memrealtime();
        pc0 = __builtin_amdgcn_s_memtime();
    }
    if constexpr (NSTG == 2) {
        for (int t = 0; t < n_t; ++t) {
            const int cur = t & 1;
            unsigned long long pa = 0;
            if constexpr (PROBE) pa = __builtin_amdgcn_s_memtime();
            sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
            __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
            if constexpr (PROBE) pwait += __builtin_amdgcn_s_memtime() - pa;
            w.next_stage = smem_p3 + (cur ^ 1) * STAGE;
            w.a_tile = smem_p3 + cur * STAGE + a_fr;
            w.w_tile = smem_p3 + cur * STAGE + w_fr;
            w.tile();
        }
    } else {
        // three stages: the DMA of k tile t + 2 is issued during tile t.  Every wave issues exactly NP DMA instructions per tile and
        // they complete in order, so "tile t has landed" is vmcnt(NP) -- __syncthreads() would drain the tile behind it as well
        w.next_stage = smem_p3 + STAGE;
        w.template pieces<0, NAG * 3 + NBW>();  // k tile 1 (or tile 0 again when there is none: dead stage)
        int cur = 0;
        for (int t = 0; t < n_t; ++t) {
            unsigned long long pa = 0;
            if constexpr (PROBE) pa = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAG * 3 + NBW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (PROBE) pwait += __builtin_amdgcn_s_memtime() - pa;
            const int nxt = cur == 0 ? 2 : cur - 1;      // (cur + 2) % 3
            w.next_stage = smem_p3 + nxt * STAGE;
            w.a_tile = smem_p3 + cur * STAGE + a_fr;
            w.w_tile = smem_p3 + cur * STAGE + w_fr;
            w.tile();
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
    // the last k tile was fetched twice (piece()); that copy must have landed before the epilogue reuses the stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long pt2 = 0, pc2 = 0;
    if constexpr (PROBE) { pt2 = __builtin_amdgcn_s_memrealtime(); pc2 = __builtin_amdgcn_s_memtime(); }

    gemm_epilogue_f32<MI, NI, WM, WN>(p, w.acc, smem_p3, m0, n0, z, lid, wave, lane, HoWo);
    if constexpr (PROBE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
        __syncthreads();
        unsigned long long* d = p.probe + 24ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.z);
        if (lane == 0) { d[4 + 2 * wave] = pc2 - pc0; d[5 + 2 * wave] = pwait; }
        if (tid == 0) { d[0] = pt0; d[1] = pt1; d[2] = pt2; d[3] = __builtin_amdgcn_s_memrealtime(); }
    }
}

template <int MI, int NI, int WM, int WN, int NSTG, bool PROBE = false>
static hipError_t launch_cfg_3p(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm3p_kernel<MI, NI, WM, WN, NSTG, PROBE>;
    constexpr int NWV = WM * WN;
    constexpr size_t stage = (size_t)(MI * WM) * 3 * 1024 + (size_t)((NI * WN * 3 + NWV - 1) / NWV) * NWV * 1024;
    // (the epilogue transposes through one 16 x (16 NI + 4) fp32 scratch per wave in the same memory)
    constexpr size_t lds = NSTG * stage > (size_t)NWV * 16 * (16 * NI + 4) * 4 ? NSTG * stage : (size_t)NWV * 16 * (16 * NI + 4) * 4;
    static_assert(lds <= 160 * 1024, "the stages must fit the CU's LDS");
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(NWV * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm3p(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesP) return hipErrorInvalidValue;
    if ((p.Cin % 32) || p.CS != 32 || !p.zero_page || !p.Bt3 || !p.A3 || p.a3_ld <= 0 || (p.a3_ld % 192) || p.out_mode != 0) return hipErrorInvalidValue;
    const bool odd_ni = (cfg == 0 || cfg == 3 || cfg == 7);
    if (p.geglu == 1 && (odd_ni || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;
    // geglu = 2 (value / gate split by wave column, combined across the two waves in the epilogue): any tile with an even number of wave columns
    if (p.geglu == 2 && (cfg == 8 || p.splits != 1 || (p.N & 3) || (p.C && (p.ldc & 3)) || (p.C3 && (p.N & 31)) || p.rowvec || p.resid || (!p.C && !p.C3))) return hipErrorInvalidValue;
    if (p.geglu != 0 && p.geglu != 1 && p.geglu != 2) return hipErrorInvalidValue;
    if ((unsigned long long)p.N * (p.geglu ? 2 : 1) * (unsigned long long)p.kt_total * 192ull >= 0xFFFFFF00ull) return hipErrorInvalidValue;   // 32-bit piece offsets
    if ((unsigned long long)p.NB * p.Hs * p.Ws * (unsigned long long)p.a3_ld >= 0xFFFFFF00ull) return hipErrorInvalidValue;
    const int bm = kTilesP[cfg].bm, bn = kTilesP[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const dim3 grid = gemm_grid(p, MT * NT);
    if (p.probe) {   // diagnostic instantiations (option gemm_probe): the three 8-wave tiles the batch-1 model uses most
        if ((unsigned long long)grid.x * grid.z > (unsigned long long)kGemmProbeBlocks) return hipErrorInvalidValue;   // the stamps would run past the buffer
        switch (cfg) {
            case 0: return launch_cfg_3p<4, 5, 4, 2, 2, true>(p, grid, stream);
            case 3: return launch_cfg_3p<2, 5, 4, 2, 2, true>(p, grid, stream);
            case 4: return launch_cfg_3p<2, 4, 4, 2, 3, true>(p, grid, stream);
        }
        return hipErrorInvalidValue;
    }
    switch (cfg) {
        case 0: return launch_cfg_3p<4, 5, 4, 2, 2>(p, grid, stream);   // 256 x 160: waves of 64 x 80
        case 1: return launch_cfg_3p<4, 4, 4, 2, 2>(p, grid, stream);   // 256 x 128: 64 x 64
        case 2: return launch_cfg_3p<4, 4, 2, 4, 2>(p, grid, stream);   // 128 x 256: 64 x 64
        case 3: return launch_cfg_3p<2, 5, 4, 2, 2>(p, grid, stream);   // 128 x 160: 32 x 80
        case 4: return launch_cfg_3p<2, 4, 4, 2, 3>(p, grid, stream);   // 128 x 128: 32 x 64, three stages
        // four-wave tiles, two or three workgroups per CU
        case 5: return launch_cfg_3p<2, 2, 2, 2, 3>(p, grid, stream);   // 64 x 64: waves of 32 x 32, 3 x 24 KB
        case 6: return launch_cfg_3p<2, 4, 2, 2, 2>(p, grid, stream);   // 64 x 128: 32 x 64, 2 x 36 KB
        case 7: return launch_cfg_3p<4, 5, 1, 4, 2>(p, grid, stream);   // 64 x 320: 64 x 80 (a whole N = 320 row per workgroup), 2 x 72 KB
        case 8: return launch_cfg_3p<2, 4, 4, 1, 2>(p, grid, stream);   // 128 x 64: 32 x 64, 2 x 36 KB
    }
    return hipErrorInvalidValue;
}

// ---- fp32 rows -> planes ----------------------------------------------------------------------------------------------------------
// x [rows][ld] fp32 (C = 32 kt channels used) -> y3 [rows][ld3 / 192 slices][3][32] bf16, slices [0, kt); chunk g of a plane row holds
// slice elements 4g..4g+3, 16+4g..16+4g+3 (s3_plane_pos).  One thread per (row, slice, chunk).  For activations that reach a plane GEMM
// as fp32 (a producer that does not write planes itself) and for the weights at load (launch_pack_split3: ld = K, ld3 = 6 K).
__global__ void split3_rows_kernel(const float* __restrict__ x, unsigned short* __restrict__ y3, long long rows, int kt, long long ld, long long ld3_elems) {
    const long long total = rows * kt * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i & 3);
        const long long rk = i >> 2;
        const long long row = rk / kt;
        const int k = (int)(rk - row * kt);
        const float* src = x + row * ld + k * 32;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src + 4 * g);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 16 + 4 * g);
        u32x4 ph, pm, plo;
        s3_split8(lo, hi, ph, pm, plo);
        unsigned short* dst = y3 + row * ld3_elems + k * 96 + g * 8;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + 32) = pm;
        *reinterpret_cast<u32x4*>(dst + 64) = plo;
    }
}

hipError_t launch_split3_rows(const float* x, void* y3, long long rows, int c, long long ld, long long ld3_bytes, hipStream_t s) {
    if ((c % 32) || (ld % 4) || (ld3_bytes % 192) || ld3_bytes < (long long)(c / 32) * 192) return hipErrorInvalidValue;
    const long long total = rows * (c / 32) * 4;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<unsigned short*>(y3), rows, c / 32, ld, ld3_bytes / 2);
    return hipGetLastError();
}

// planes -> fp32 rows: x = (h + m) + l exactly (h + m has at most 16 significant bits).  Test / debugging aid: the operator-level entry
// points use it to hand a plane producer's result back in the reference's layout.
__global__ void join3_rows_kernel(const unsigned short* __restrict__ x3, float* __restrict__ y, long long rows, int kt, long long ld3_elems, long long ld) {
    const long long total = rows * kt * 32;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 31);
        const long long rk = i >> 5;
        const long long row = rk / kt;
        const int k = (int)(rk - row * kt);
        const unsigned short* src = x3 + row * ld3_elems + k * 96 + s3_plane_pos(j);
        const float h = __builtin_bit_cast(float, (unsigned)src[0] << 16);
        const float m = __builtin_bit_cast(float, (unsigned)src[32] << 16);
        const float l = __builtin_bit_cast(float, (unsigned)src[64] << 16);
        y[row * ld + k * 32 + j] = (h + m) + l;
    }
}

hipError_t launch_join3_rows(const void* x3, float* y, long long rows, int c, long long ld3_bytes, long long ld, hipStream_t s) {
    if ((c % 32) || (ld3_bytes % 192)) return hipErrorInvalidValue;
    const long long total = rows * c;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(join3_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(x3), y, rows, c / 32, ld3_bytes / 2, ld);
    return hipGetLastError();
}

}  // namespace sdmi
