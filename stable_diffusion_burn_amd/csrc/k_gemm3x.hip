// k_gemm3x.hip -- fp32 implicit-GEMM conv / linear on the bf16 matrix pipe ("split" kernel, precision = 0).
//
// v_mfma_f32_16x16x4_f32 retires 256 flop/clk/CU; v_mfma_f32_16x16x32_bf16 4096.  An fp32 number is EXACTLY the sum of
// three bf16 numbers (round-to-nearest splits: x = h + m + l, |m| <= 2^-9 |x|, |l| <= 2^-17 |x|; 8 + 8 + 8 significand
// bits, same exponent range as fp32), the product of two bf16 numbers is exact in fp32, and the MFMA accumulates in fp32.
// So a*w = (ah + am + al)(wh + wm + wl) is accumulated as the six partial products that are not below 2^-24 |a w|:
//      wl*ah, wh*al, wm*am, wm*ah, wh*am, wh*ah          (dropped: wm*al + wl*am + wl*al)
// The dropped terms are at most 2^-23 |a w| (both operands just above a power of two: about one fp32 ulp of the product), 2^-28 |a w|
// on average -- tests/test_split_oracle_cpu.py proves this on the numpy restatement oracle/split_oracle.py -- and the sum is an fp32
// sum as in the fp32 MFMA, whose every accumulation step rounds away up to 2^-24.  Six bf16 MFMAs per 16x16x32 block = 96 matrix-pipe cycles against 256 for the eight
// 16x16x4 fp32 MFMAs of the same block: the matrix-pipe bound of an fp32 GEMM moves from 157 to 417 TFLOP/s.  Same parity
// bars as k_gemm2x.hip (tests/test_ops_gpu.py); DESIGN.md section 4a has the error analysis and the measurements.
//
// Structure: k_gemm2x.hip (8 waves, LDS-DMA staged, double-buffered 32-channel k tiles, XCD-aware tile map, deterministic
// split-K slabs, shared epilogue) with
//   * activations staged as fp32 exactly as there and split IN REGISTERS after the fragment read (v_cvt_pk_bf16_f32, shift / and,
//     v_sub_f32: 44 VALU instructions per 16x32 fragment -- 36 with v_pk_add_f32, variant bit 1 --, issued one or two at a time
//     between the 6 NI MFMAs of the previous fragment);
//   * weights split once at load into three bf16 planes (launch_pack_split3): [N][kt][plane][32] bf16, 192 bytes per row per
//     k tile, staged as 16-row x 64-byte pieces (one DMA instruction = one plane of one 16-row fragment group) and read as
//     one ds_read_b128 per plane per fragment.
// The 128-row tiles (32-row wave tiles: a k tile is 0.7 us of matrix work, less than the latency of the DMA behind it) run
// three LDS stages with a hand-counted vmcnt instead of two with __syncthreads(): +5-10 % on long-K shapes.
// k order inside a k tile: lane group g = lane >> 4 of the MFMA supplies k = {4g..4g+3, 16+4g..16+4g+3} -- the two 16-byte
// chunks (g, 4 + g) of the fp32 row that k_gemm2x.hip's conflict-free fragment reads fetch; the pack kernel stores the
// weight planes in that order (chunk g of a plane = those 8 elements).
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_epi.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

static const GemmTileInfo kTilesS[kNumGemmTilesS] = {
    {256, 160, "256x160s"}, {128, 320, "128x320s"}, {256, 128, "256x128s"}, {128, 256, "128x256s"}, {128, 160, "128x160s"}, {128, 128, "128x128s"}};
const GemmTileInfo& gemm_tile_info_s(int cfg) { return kTilesS[cfg]; }

// Per-wave state of the k loop.  Everything is indexed with compile-time constants (member templates), so the arrays live in
// registers; the issue order of one k tile is spelled out instruction group by instruction group and fenced with
// sched_barrier(0), because (a) hipcc otherwise hoists all the splits in front of the MFMAs and (b) with LDS-DMA in flight its
// own waits are all lgkmcnt(0), so a fragment read must be issued well before, and never right in front of, a first use.
// (Pipelined forms of this loop -- the tile head hoisted into the previous tile's last row, weight planes prefetched into dead registers, inline-asm reads
// with hand-counted lgkmcnt waits -- were built, verified and measured no faster in round 2 (profiles/r02y_*, r02zz_*) and removed.)
template <int MI, int NI, int NA, int NBW, int PW, int A_BYTES, bool SPREAD, bool SCALAR, int NSTG = 2>
struct S3Wave {
    using S3Split = S3SplitT<SCALAR>;
    static constexpr int kS3Steps = S3Split::kSteps;
    static constexpr int NMF = 6 * NI;        // MFMAs of one fragment row
    static constexpr int NP = NA + NBW;       // DMA pieces of one k tile, issued between the MFMAs of row 0 (and 1)
    static constexpr int DMA_ROWS = (MI > 1) ? 2 : 1;

    f32x4 acc[MI][NI];
    u32x4 wf[3][NI];
    S3Split sp[2];
    f32x4 raw[2][2];
    // DMA sources
    int a_iy0[NA], a_ix0[NA];
    unsigned a_off[NA];               // byte offset of the sample + this lane's chunk (operands are < 4 GiB: launch_gemm checks)
    unsigned w_off[NBW];
    unsigned w_kstep;           // bytes between a weight piece's k tiles: 192 (row-major planes) or 3072 (16-row groups, ConvGemm::b3_grouped)
    const char *Abase, *Wbase, *zero;
    unsigned pix_bytes;
    int Hin, Win, ups, Ws, KH, KW, wave;
    int cs, ky, kx, kt_next, kt_end;
    // current k tile
    const unsigned char* a_tile;      // stage + this wave's activation rows
    const unsigned char* w_tile;      // stage + this wave's weight pieces + lane offset
    unsigned char* next_stage;        // where the DMA of k tile kt_next goes
    int fr_off0, fr_off1;

    // DMA piece J of k tile kt_next -> next_stage.  Straight-line code (selects, no branches: a branch between the MFMAs lets
    // hipcc sink the split arithmetic out of the slots it is placed in); the LDS stage has room for 8 NBW weight pieces, the
    // pieces past PW re-fetch the last row.  After the last piece the source moves on to the next k tile, unless there is none:
    // then the same tile is fetched once more into the stage nobody reads any more.
    template <int J>
    __device__ __forceinline__ void piece() {
        if constexpr (J < NA) {
            const int iy = a_iy0[J] + ky;
            const int ix = a_ix0[J] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const unsigned off = a_off[J] + (unsigned)((iy >> ups) * Ws + (ix >> ups)) * pix_bytes + (unsigned)cs * 128u;
            const char* src = (ok ? Abase : zero) + (ok ? off : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(next_stage + (wave + 8 * J) * 1024), 16, 0, 0);
        } else {
            constexpr int j = J - NA;
            const char* src = Wbase + (w_off[j] + (unsigned)kt_next * w_kstep);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(next_stage + A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
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

    template <int F>
    __device__ __forceinline__ void read_fragment() {     // activation fragment F of the wave's rows -> raw[F & 1]
        raw[F & 1][0] = *reinterpret_cast<const f32x4*>(a_tile + F * 2048 + fr_off0);
        raw[F & 1][1] = *reinterpret_cast<const f32x4*>(a_tile + F * 2048 + fr_off1);
    }

    // MFMA K of fragment row MIDX: partial product K / NI (smallest first: wl*ah, wh*al, wm*am, wm*ah, wh*am, wh*ah) on column
    // fragment K % NI; behind it this slot's share of the next fragment's split, of the next k tile's DMA, and -- half a
    // row ahead of its first use -- the read of fragment MIDX + 2
    template <int MIDX, int K>
    __device__ __forceinline__ void mfmas() {
        if constexpr (K < NMF) {
            constexpr int WP[6] = {2, 0, 1, 1, 0, 0};
            constexpr int pr = K / NI, ni = K % NI, b = MIDX & 1;
            const u32x4& a = (pr == 0 || pr == 3 || pr == 5) ? sp[b].h : ((pr == 1) ? sp[b].l : sp[b].m);
            acc[MIDX][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[WP[pr]][ni]), __builtin_bit_cast(bf16x8, a),
                                                                    acc[MIDX][ni], 0, 0, 0);
            if constexpr (MIDX + 1 < MI) sp[b ^ 1].template steps<K * kS3Steps / NMF, (K + 1) * kS3Steps / NMF>();
            if constexpr (SPREAD && MIDX < DMA_ROWS) {
                constexpr int slot = MIDX * NMF + K, slots = DMA_ROWS * NMF;
                pieces<slot * NP / slots, (slot + 1) * NP / slots>();
            }
            if constexpr (MIDX + 2 < MI && K == NMF / 2) read_fragment<MIDX + 2>();
            __builtin_amdgcn_sched_barrier(0);
            mfmas<MIDX, K + 1>();
        }
    }
    template <int MIDX>
    __device__ __forceinline__ void rows() {
        if constexpr (MIDX < MI) {
            if constexpr (MIDX + 1 < MI) sp[(MIDX & 1) ^ 1].load(raw[(MIDX & 1) ^ 1][0], raw[(MIDX & 1) ^ 1][1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas<MIDX, 0>();
            rows<MIDX + 1>();
        }
    }

    // one k tile: the two first activation fragments, the head of split 0 (the wait lands here, with only those four reads
    // outstanding), the weight planes (in flight during the rest of split 0), then the fragment rows
    __device__ __forceinline__ void tile() {
        if constexpr (!SPREAD) pieces<0, NP>();
        read_fragment<0>();
        if constexpr (MI > 1) read_fragment<1>();
        __builtin_amdgcn_sched_barrier(0);
        sp[0].load(raw[0][0], raw[0][1]);
        sp[0].template steps<0, 4>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[pl][ni] = *reinterpret_cast<const u32x4*>(w_tile + (ni * 3 + pl) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        sp[0].template steps<4, kS3Steps>();
        __builtin_amdgcn_sched_barrier(0);
        rows<0>();
    }
};

template <int MI, int NI, int WM, int WN, bool SPREAD, bool SCALAR, int NSTG>
__global__ __launch_bounds__(512) void conv_gemm3x_kernel(const ConvGemm p) {
    static_assert(NSTG == 2 || NSTG == 3, "LDS stages");
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM % 64 == 0, "every wave issues whole 8-row DMA pieces of the activation tile");
    constexpr int NA = BM / 64;               // activation pieces (8 rows x 128 B) per wave per k tile
    constexpr int PW = (BN / 16) * 3;         // weight pieces (16 rows x 64 B: one plane of one fragment group) per k tile
    constexpr int NBW = (PW + 7) / 8;         // ... per wave (the last one only on waves < PW % 8)
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE = A_BYTES + NBW * 8 * 1024;   // bytes of one LDS stage (weights: BN * 192, rounded up to 8 pieces per wave)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x32[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    // GEGLU mode: as k_gemm2x.hip (fragment ni even = values, odd = gates of the same outputs)
    constexpr int WNC = 16 * NI;
    const bool geglu = p.geglu != 0;
    const int BNO = geglu ? BN / 2 : BN;
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

    S3Wave<MI, NI, NA, NBW, PW, A_BYTES, SPREAD, SCALAR, NSTG> w;
    w.Hin = p.Hs << p.ups;
    w.Win = p.Ws << p.ups;
    w.ups = p.ups;
    w.Ws = p.Ws;
    w.KH = p.KH;
    w.KW = p.KW;
    w.wave = wave;
    w.pix_bytes = (unsigned)p.a_ld * 4u;
    w.Abase = reinterpret_cast<const char*>(p.A);
    w.Wbase = reinterpret_cast<const char*>(p.Bt3);
    w.zero = reinterpret_cast<const char*>(p.zero_page);

    // activation piece j of a wave: tile rows (wave + 8 j) * 8 .. + 7; lane -> row + (lane >> 3), LDS slot lane & 7 receives
    // global chunk (lane & 7) ^ (row & 7)
    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ sub;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave + 8 * j) * 8 + sub;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        w.a_off[j] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * w.pix_bytes + chunk * 16;
        w.a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
        w.a_ix0[j] = ox * p.stride - p.pad;
    }
    // weight piece q = wave + 8 j: fragment group q / 3, plane q % 3; lane -> row lane >> 2, LDS slot lane & 3 receives the
    // plane's 16-byte chunk (lane & 3) ^ f(row), f(r) = (-(r >> 2)) & 3  (conflict-free ds_read_b128 of 64-byte rows: the
    // instruction's 16-lane groups {0-3, 12-15, 20-27}, ... then cover 16 distinct slots of the 256-byte bank window)
    const unsigned w_row_bytes = (unsigned)p.kt_total * 192u;
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int q = wave + 8 * j;
        const int f = q / 3, pl = q - 3 * f;
        const int r = lane >> 2;
        const int ch = (lane & 3) ^ ((-(r >> 2)) & 3);
        const int r0 = f * 16 + r;             // tile row of the weight operand
        int n = n0 + r0;
        long long wrow = n;
        if (geglu) {
            const int fw = f / NI, ni = f - fw * NI;
            n = n0 + fw * (WNC / 2) + (ni >> 1) * 16 + r;
            wrow = (long long)n + ((ni & 1) ? p.N : 0);
        }
        // rows past N (ragged last tile) and the pieces past PW fetch the last valid row instead: real memory, and the
        // accumulator columns they feed are never stored
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

    // fragment reads: activations as k_gemm2x.hip (row c, chunks g and 4 + g, XOR (c & 7)); weights row c of a piece, slot g ^ f(c)
    const int c15 = lane & 15, g4 = lane >> 4;
    w.fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    w.fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
    const int a_base = wm * 16 * MI * 128;
    const int w_fr = A_BYTES + wn * NI * 3 * 1024 + c15 * 64 + ((g4 ^ ((-(c15 >> 2)) & 3)) << 4);

#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) w.acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // variant bit 4: static priority for the second-dispatched half of the workgroup, the arbitration loser of every SIMD's pair (measured: no gain)
    if ((p.variant & 16) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    w.next_stage = smem_x32;
    w.template pieces<0, NA + NBW>();      // k tile 0
    if constexpr (NSTG == 2) {
        for (int t = 0; t < n_t; ++t) {
            const int cur = t & 1;
            sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
            __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
            w.next_stage = smem_x32 + (cur ^ 1) * STAGE;
            w.a_tile = smem_x32 + cur * STAGE + a_base;
            w.w_tile = smem_x32 + cur * STAGE + w_fr;
            w.tile();
        }
    } else {
        // Three stages: the DMA of k tile t + 2 is issued during tile t, so a tile's data has two tile times to arrive -- a k tile
        // of a 32-row wave tile (0.7 us) is shorter than the latency of the DMA it would otherwise wait for.  Every wave issues
        // exactly NA + NBW DMA instructions per tile (piece() has no branches) and they complete in order, so "tile t has
        // landed" is vmcnt(NA + NBW): hand-written, because __syncthreads() would drain the tile behind it as well.
        w.next_stage = smem_x32 + STAGE;
        w.template pieces<0, NA + NBW>();      // k tile 1 (or tile 0 again when there is none: dead stage)
        int cur = 0;
        for (int t = 0; t < n_t; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NBW) : "memory");
            __builtin_amdgcn_s_barrier();       // k tile t is in LDS; every wave is done with tile t - 1, whose stage tile t + 2 takes
            asm volatile("" ::: "memory");
            const int nxt = cur == 0 ? 2 : cur - 1;      // (cur + 2) % 3
            w.next_stage = smem_x32 + nxt * STAGE;
            w.a_tile = smem_x32 + cur * STAGE + a_base;
            w.w_tile = smem_x32 + cur * STAGE + w_fr;
            w.tile();
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
    // the last k tile was fetched twice (piece()); that copy must have landed before the epilogue reuses the stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.variant & 16) __builtin_amdgcn_s_setprio(0);

    gemm_epilogue_f32<MI, NI, WM, WN>(p, w.acc, smem_x32, m0, n0, z, lid, wave, lane, HoWo);
}

template <int MI, int NI, int WM, int WN, bool SPREAD, bool SCALAR, int NSTG = 2>
static hipError_t launch_cfg_3x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm3x_kernel<MI, NI, WM, WN, SPREAD, SCALAR, NSTG>;
    constexpr size_t lds = NSTG * ((size_t)(16 * MI * WM) * 128 + (size_t)((NI * WN * 3 + 7) / 8) * 8192);
    static_assert(lds <= 160 * 1024, "the stages must fit the CU's LDS");
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm3x(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesS) return hipErrorInvalidValue;
    if ((p.Cin % 32) || p.CS != 32 || !p.zero_page || !p.Bt3 || p.out_mode != 0) return hipErrorInvalidValue;
    const bool odd_ni = (cfg == 0 || cfg == 1 || cfg == 4);
    if (p.geglu != 0 && p.geglu != 1) return hipErrorInvalidValue;   // the wave-column form (geglu = 2) exists in k_gemm3p.hip only: this kernel's weight-piece mapping is the interleaved form
    if (p.geglu && (odd_ni || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;
    if ((unsigned long long)p.N * (p.geglu ? 2 : 1) * (unsigned long long)p.kt_total * 192ull >= 0xFFFFFF00ull) return hipErrorInvalidValue;   // 32-bit piece offsets
    const int bm = kTilesS[cfg].bm, bn = kTilesS[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const int tiles = MT * NT;
    const dim3 grid = gemm_grid(p, tiles);
    // p.variant bit 0: issue the next k tile's DMA in one block behind the barrier instead of between the MFMAs of rows 0 / 1
    // bit 1: the split's residual subtractions as scalar v_sub_f32 pairs instead of v_pk_add_f32
    const bool spread = !(p.variant & 1), scalar = (p.variant & 2) != 0;
#define SDMI_3X(MI, NI, WM, WN)                                                                                       \
    (spread ? (scalar ? launch_cfg_3x<MI, NI, WM, WN, true, true>(p, grid, stream) : launch_cfg_3x<MI, NI, WM, WN, true, false>(p, grid, stream)) \
            : (scalar ? launch_cfg_3x<MI, NI, WM, WN, false, true>(p, grid, stream) : launch_cfg_3x<MI, NI, WM, WN, false, false>(p, grid, stream)))
    switch (cfg) {
        case 0: return SDMI_3X(4, 5, 4, 2);
        case 1: return SDMI_3X(4, 5, 2, 4);
        case 2: return SDMI_3X(4, 4, 4, 2);
        case 3: return SDMI_3X(4, 4, 2, 4);
        // the 32-row wave tiles: three LDS stages unless variant bit 2 is set (DMA of tile t + 2 in flight during tile t)
        case 4: return (p.variant & 4) ? SDMI_3X(2, 5, 4, 2) : (scalar ? launch_cfg_3x<2, 5, 4, 2, true, true, 3>(p, grid, stream) : launch_cfg_3x<2, 5, 4, 2, true, false, 3>(p, grid, stream));
        case 5: return (p.variant & 4) ? SDMI_3X(2, 4, 4, 2) : (scalar ? launch_cfg_3x<2, 4, 4, 2, true, true, 3>(p, grid, stream) : launch_cfg_3x<2, 4, 4, 2, true, false, 3>(p, grid, stream));
    }
#undef SDMI_3X
    return hipErrorInvalidValue;
}

// ---- weight planes ---------------------------------------------------------------------------------------------------
// bt [rows][K] fp32 in the kernels' k order (K % 32 == 0) -> w3 [rows][K / 32][3][32] bf16: split3_rows_kernel (k_gemm3p.hip), the
// split every plane producer uses (k_split3.hpp: chunk g of a plane row = k-tile elements 4g..4g+3, 16+4g..16+4g+3).
// grouped (round 5; rows % 16 == 0): [rows / 16][K / 32][3][16][32] -- the 16 rows of a fragment group side by side per (k tile, plane), so that the DMA piece a wave
// fetches (16 rows x 64 B of one plane) is 1 KiB of consecutive bytes and a group's k tiles are consecutive 3 KiB blocks: the weight stream of a small-M layer (8 x 8 and
// 16 x 16 levels: 88-176 MB of planes read once per forward) becomes sequential per fragment group instead of 64-byte pieces 6 K bytes apart.
__global__ void pack_split3_grouped_kernel(const float* __restrict__ x, unsigned short* __restrict__ y3, long long rows, int kt) {
    const long long total = rows * kt * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i & 3);
        const long long rk = i >> 2;
        const long long row = rk / kt;
        const int k = (int)(rk - row * kt);
        const float* src = x + row * ((long long)kt * 32) + k * 32;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src + 4 * g);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 16 + 4 * g);
        u32x4 ph, pm, plo;
        s3_split8(lo, hi, ph, pm, plo);
        unsigned short* dst = y3 + ((row >> 4) * kt + k) * 1536 + (row & 15) * 32 + g * 8;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + 512) = pm;
        *reinterpret_cast<u32x4*>(dst + 1024) = plo;
    }
}

hipError_t launch_pack_split3(const float* bt, void* w3, long long rows, int K, hipStream_t s, bool grouped) {
    if (K % 32) return hipErrorInvalidValue;
    if (!grouped) return launch_split3_rows(bt, w3, rows, K, K, (long long)(K / 32) * 192, s);
    if (rows % 16) return hipErrorInvalidValue;
    const long long total = rows * (K / 32) * 4;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pack_split3_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bt, reinterpret_cast<unsigned short*>(w3), rows, K / 32);
    return hipGetLastError();
}

}  // namespace sdmi
