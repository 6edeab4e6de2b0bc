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
// Round 6, measured and removed (tools/probes/r06w_k_gemm_bf16r_probe.hip.txt, profiles/r06w_ring_vs_two_stage.txt): a FOUR-stage ring of 32-deep k steps on the 256 x 256 /
// 256 x 128 tiles (DMA three steps ahead, counted vmcnt) -- bit-identical (72 operator cases incl. split-K) and 10 - 12 % SLOWER on every shape, hot and cold: the
// 30 % of wave cycles these kernels spend at the barrier (SQ_WAIT_ANY, profiles/r06v_*) is not the DMA round trip, and twice the barriers cost more than the lead time buys.
// k order, weight packing, swapped MFMA operands (a lane holds 4 consecutive output channels), XCD-aware tile
// map, deterministic split-K and the fused epilogue are those of k_gemm_bf16.hip.
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_gemm_bf16_epi.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

static const GemmTileInfo kTilesX[kNumGemmTilesX] = {
    {256, 320, "256x320x"}, {256, 256, "256x256x"}, {256, 128, "256x128x"}, {128, 320, "128x320x"}};
const GemmTileInfo& gemm_tile_info_x(int cfg) { return kTilesX[cfg]; }

// PERSIST (round 5; ConvGemm::variant bit 0, launches without split-K and with more tiles than workgroups): the grid is one workgroup per CU and every workgroup
// walks the tiles vb = blockIdx.x, blockIdx.x + gridDim.x, ... of the same XCD-aware order.  Behind the last k tile of a tile: workgroup barrier (every wave is done
// with stage L) -> the FIRST k tile of the next tile is DMA'd into stage L -> epilogue with its LDS scratch in stage L ^ 1 (free since the top of the last k
// iteration) -> the next tile's k loop starts on stage L.  What a tile's launch + prologue cost (profiles/r04ad: 2.2 us of a short-K tile's 11 us of fixed cost) hides
// behind the epilogue, and the epilogue's stores drain behind the next tile's k loop instead of in front of the next workgroup's start.  Same products in the same
// order: bit-identical to the one-tile-per-workgroup form.
// PM: -1 = one tile per workgroup; 0 / 2 = persistent, with the epilogue's mode fixed at compile time (k_gemm_bf16_epi.hpp: plain / GEGLU gate; bf16 out, no residual).
// In the tile loop hipcc hoists every lane-derived invariant of the per-tile code (piece rows, scratch offsets, ...) in front of the loop, where it lives through the k
// loop beside 160 accumulators and 52 fragment registers -- 30 to 200 spilled registers; the per-tile code therefore derives them from an opaque copy of the lane index.
template <int MI, int NI, int WM, int WN, int PM, bool LIN = false>
__global__ __launch_bounds__(512) void conv_gemm_bf16x_kernel(const ConvGemm p) {
    constexpr bool PERSIST = PM >= 0;
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
    const bool geglu = PERSIST ? PM == 2 : p.geglu != 0;
    const int BNO = geglu ? BN / 2 : BN;   // output columns per tile
    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BNO - 1) / BNO;
    // tile vb of the XCD-aware order (gemm_work_of_block with the grid's x extent spelled out, so that it also holds for vb >= gridDim.x)
    const int tiles = MT * NT;
    const int tpx = (tiles + 7) >> 3;
    int vb = blockIdx.x;
    auto tile_of = [&](int b, int& tm_, int& tn_) {
        const int j = b >> 3;
        const int lid = (b & 7) * tpx + j;
        tm_ = lid / NT;
        tn_ = lid - tm_ * NT;
        return j < tpx && lid < tiles;
    };
    int tm, tn;
    if (!tile_of(vb, tm, tn)) return;
    int m0 = tm * BM;
    int n0 = tn * BNO;

    const int z = PERSIST ? 0 : (int)blockIdx.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    // Linear layers and 1x1 convolutions (round 6): output row m IS input pixel m -- no sample / row / column split of m (two integer divisions per DMA piece, in setup()
    // and again in the tile loop's issue_first(): profiles/r06zg_*, 1.9 us of a 256 x 256 tile's 20.6 us at K = 320)
    // The tile loop takes the case as a template parameter (LIN, chosen by the launcher): as a run-time flag both address forms sit in its per-tile code and the
    // 256 x 320 tile spills.
    auto is_lin = [&]() {
        if constexpr (PERSIST) return LIN;
        return T == 1 && p.stride == 1 && p.pad == 0 && p.ups == 0 && p.Ho == p.Hs && p.Wo == p.Ws;
    };
    f32x4 acc[MI][NI];
    const int c15 = lane & 15, g4 = lane >> 4;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const long long pix_bytes = (long long)p.a_ld * 2;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // DMA piece j of a wave covers tile rows (wave + 8 j) * 8 .. + 7; lane -> row + (lane >> 3) =: sub, LDS slot lane & 7,
    // which receives global chunk (lane & 7) ^ (row & 7) = (lane & 7) ^ (lane >> 3)
    int a_iy0[NA], a_ix0[NA];
    long long a_nboff[NA];
    const char* b_src[NB];
    int cs, ky, kx, kt_next;
    auto opaque_lane = [&]() {
        int l = lane;
        if constexpr (PERSIST) asm volatile("" : "+v"(l));
        return l;
    };
    auto setup = [&]() {   // source addresses of tile (m0, n0); rewinds the k walk
        const int lane_o = opaque_lane();
        const int sub = lane_o >> 3;
        const int chunk = (lane_o & 7) ^ sub;
        const bool lin = is_lin();
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m0 + (wave + 8 * j) * 8 + sub;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            if (lin) {
                a_nboff[j] = (long long)mm * pix_bytes + chunk * 16;
                a_iy0[j] = ok ? 0 : -(1 << 28);
                a_ix0[j] = 0;
                continue;
            }
            const int nb = mm / HoWo;
            const int rem = mm - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            a_nboff[j] = (long long)nb * (p.Hs * p.Ws) * pix_bytes + chunk * 16;
            a_iy0[j] = ok ? oy * p.stride - p.pad : -(1 << 28);   // rows past M: never in range -> zero page
            a_ix0[j] = ox * p.stride - p.pad;
        }
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
        cs = kt_begin / T;
        const int tap0 = kt_begin - cs * T;
        ky = tap0 / p.KW;
        kx = tap0 - ky * p.KW;
        kt_next = kt_begin;
    };

    auto advance_k = [&]() {
        const bool wrap_x = (kx + 1 == p.KW);
        const bool wrap_y = wrap_x && (ky + 1 == p.KH);
        kx = wrap_x ? 0 : kx + 1;
        ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
        cs = wrap_y ? cs + 1 : cs;
        ++kt_next;
    };
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
        advance_k();
    };

    // persistent form: the first k tile (channel slice 0, tap 0) of tile (m0n, n0n) into stage buf, every piece's address computed right in front of its DMA and not
    // kept -- this runs between a tile's k loop and its epilogue, beside 160 live accumulators
    auto issue_first = [&](int buf, int m0n, int n0n) {
        unsigned char* stage = smem_x + buf * STAGE;
        const int lane_o = opaque_lane();
        const int sub = lane_o >> 3;
        const int chunk = (lane_o & 7) ^ sub;
        const bool lin = is_lin();
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m0n + (wave + 8 * j) * 8 + sub;
            const int mm = m < p.M ? m : 0;
            if (lin) {
                const char* src = m < p.M ? Abase + (long long)mm * pix_bytes + chunk * 16 : zero;
                __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
                continue;
            }
            const int nb = mm / HoWo;
            const int rem = mm - nb * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy = oy * p.stride - p.pad, ix = ox * p.stride - p.pad;
            const bool ok = (m < p.M) & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
            const long long pix = (long long)((iy >> p.ups) * p.Ws + (ix >> p.ups));
            const char* src = ok ? Abase + (long long)nb * (p.Hs * p.Ws) * pix_bytes + chunk * 16 + pix * pix_bytes : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r0 = (wave + 8 * j) * 8 + sub;
            int n = n0n + r0;
            long long wrow = n;
            if (geglu) {
                const int f = r0 >> 4, fw = f / NI, ni = f - fw * NI;
                n = n0n + fw * (WNC / 2) + (ni >> 1) * 16 + (r0 & 15);
                wrow = (long long)n + ((ni & 1) ? p.N : 0);
            }
            const char* src = (n < p.N) ? Bbase + wrow * p.b_ld * 2 + chunk * 16 : zero;
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(stage + BM * 128 + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };

    // fragment reads: lane (c = lane & 15, g = lane >> 4) reads row base + c, chunk (4 kk + g) ^ (c & 7)
    const int fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    const int fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
    const int a_base = wm * 16 * MI * 128;
    const int b_base = BM * 128 + wn * 16 * NI * 128;

    setup();
    int s0 = 0;      // LDS stage of the tile's first k tile
    issue(0);
    constexpr int GT = PM < 0 ? -1 : PM == 2 ? 1 : 0;
    // the per-column part of the accumulators' initial value (k_gemm_bf16_epi.hpp): the tile loop requests the NEXT tile's in front of the epilogue's stores (vmcnt retires in
    // order: behind them the loads waited out every store of the tile, 2.1 us per tile -- profiles/r06zh_*).  The 256 x 320 tile has no registers for the 20 values beside
    // its epilogue (68 spilled) and loads them where it always did.
    constexpr bool COLS_AHEAD = PERSIST && MI * NI <= 32;
    bepi_f32x4 colv[NI];
    if constexpr (COLS_AHEAD) gemm_acc_cols_bf16<NI, WN, GT>(p, colv, n0, wave, opaque_lane());
    for (;;) {
    // the bias, or zero; + the residual tile (ConvGemm::resid_acc)
    constexpr int RBM = PERSIST ? 2 : NI > 4 ? 2 : 4;      // fragment rows of residual loads in flight: what the registers beside the tile loop allow
    if constexpr (COLS_AHEAD) gemm_acc_init_bf16<MI, NI, WM, WN, GT, RBM>(p, acc, colv, m0, n0, wave, opaque_lane(), HoWo);
    else gemm_acc_init_bf16<MI, NI, WM, WN, GT, false, RBM>(p, acc, m0, n0, wave, PERSIST ? opaque_lane() : lane, HoWo);
    for (int t = 0; t < n_t; ++t) {
        const int cur = (s0 + t) & 1;
        sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        // variant bit 2 (round 6, default): the second wave of every SIMD (waves 4 - 7) issues its DMA pieces BETWEEN the tile's two k steps instead of in front of them, so that
        // behind the barrier half of the waves read fragments and start the matrix pipe while the other half issue DMA (every wave used to do both in the same order at
        // the same time: an LDS read burst of 72 KB with an idle matrix pipe at the top of every k tile).  One-tile forms only: the tile loop has no registers for the
        // second issue point (tried: 19 - 27 SGPRs spilled to lanes, 31 VGPRs on the 256 x 320 tile; on the 256 x 256 / 256 x 128 tile loops no gain per image, profiles/r06za_*).
        // Same products in the same order: bit-identical.  Per shape -5 ... -10 % (profiles/r06x_*, r06y_*).
        const bool late_dma = !PERSIST && (p.variant & 4) && wave >= 4;     // (the one-tile 256 x 256 / 256 x 128 forms: the 256 x 320 tile and the tile loop have no registers for the second issue point)
        if (t + 1 < n_t && !late_dma) issue(cur ^ 1);
        const unsigned char* stage = smem_x + cur * STAGE;
        // Fragment reads run one group of rows ahead of the MFMAs that use them: the reads of rows [g+1] are issued
        // before the MFMAs of rows [g], so the compiler's counted lgkmcnt waits find the data already there
        // (one ds_read_b128 per 5 MFMAs; issuing them two at a time right before use stalls every 10 MFMAs).
        constexpr int GM = (MI >= 8) ? 4 : 2;    // rows per group
        constexpr int NG = MI / GM;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1 && t + 1 < n_t && late_dma) issue(cur ^ 1);
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

    if constexpr (!PERSIST) {
        gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_x, m0, n0, z, wave, lane, HoWo);
        break;
    } else {
        const int L = (s0 + n_t - 1) & 1;     // stage of the last k tile
        const int m0c = m0, n0c = n0;
        vb += gridDim.x;
        const bool more = tile_of(vb, tm, tn);
        __syncthreads();                      // every wave is done with stage L (and, since the top of the last k iteration, with stage L ^ 1)
        if (more) {
            m0 = tm * BM;
            n0 = tn * BNO;
            issue_first(L, m0, n0);
            if constexpr (COLS_AHEAD) gemm_acc_cols_bf16<NI, WN, GT>(p, colv, n0, wave, opaque_lane());
        }
        // (the lane index is made opaque per tile: the epilogue's lane-derived offsets are then recomputed here instead of being hoisted out of the tile loop, where
        // they would live through the k loop beside the accumulators)
        gemm_epilogue_bf16<MI, NI, WM, WN, PM>(p, acc, smem_x + (L ^ 1) * STAGE, m0c, n0c, 0, wave, opaque_lane(), HoWo, true);
        if (!more) break;
        s0 = L;
        setup();          // (the addresses issue_first computed were not kept: 26 registers that would live through the epilogue)
        advance_k();      // (k tile 0 is on its way)
    }
    }
}

template <int MI, int NI, int WM, int WN, int PM, bool LIN = false>
static hipError_t launch_cfg_bf16x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16x_kernel<MI, NI, WM, WN, PM, LIN>;
    constexpr size_t lds = 2 * (size_t)(16 * MI * WM + 16 * NI * WN) * 128;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}
template <int MI, int NI, int WM, int WN>
static hipError_t launch_persist_bf16x(const ConvGemm& p, int mode, dim3 grid, hipStream_t stream) {
    const bool lin = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0 && p.ups == 0 && p.Ho == p.Hs && p.Wo == p.Ws;   // (kernel: LIN)
    if (mode == 0) return lin ? launch_cfg_bf16x<MI, NI, WM, WN, 0, true>(p, grid, stream) : launch_cfg_bf16x<MI, NI, WM, WN, 0>(p, grid, stream);
    if constexpr (NI % 2 == 0) {
        if (mode == 2) return lin ? launch_cfg_bf16x<MI, NI, WM, WN, 2, true>(p, grid, stream) : launch_cfg_bf16x<MI, NI, WM, WN, 2>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

// workgroups of a persistent launch: one per CU of the device the stream belongs to (these tiles take the whole LDS, so more would only queue)
static int persistent_workgroups() {
    static int n_cu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n_cu[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = (v / 8) * 8 > 0 ? (v / 8) * 8 : 8;    // a multiple of the 8 XCDs: workgroup b stays on XCD b % 8 for every tile it walks
    }
    return n_cu[dev];
}

// the persistent kernel's epilogue mode for this launch, or -1: ConvGemm::variant bit 0, no split-K, more tiles than workgroups, no residual (that path of the
// epilogue spills inside the tile loop) and the conditions under which gemm_epilogue_bf16 takes its 16-byte bf16 paths (checked here, compiled in there)
int conv_gemm_bf16x_persistent_mode(const ConvGemm& p, int cfg) {
    if (!(p.variant & 1) || p.splits != 1 || cfg < 0 || cfg >= kNumGemmTilesX || p.out_mode == 1) return -1;
    // (round 6, measured and not kept -- profiles/r06z_*: sending launches of >= 8 k tiles to the staggered one-tile form instead of the tile loop: GEMM class 30.94 -> 31.10 ms)
    if ((p.N & 7) || (p.ldc & 7) || (p.resid && !(p.resid_acc & 1))) return -1;
    const int bm = kTilesX[cfg].bm, bn = kTilesX[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bno - 1) / bno);
    if (tiles <= persistent_workgroups()) return -1;
    return p.geglu ? 2 : 0;
}

hipError_t launch_conv_gemm_bf16x(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesX) return hipErrorInvalidValue;
    if ((p.Cin % 64) || !p.zero_page) return hipErrorInvalidValue;
    if (p.geglu && (cfg == 0 || cfg == 3 || p.splits != 1 || (p.N & 7) || (p.ldc & 7) || p.rowvec || p.resid)) return hipErrorInvalidValue;  // needs an even NI
    const int bm = kTilesX[cfg].bm, bn = kTilesX[cfg].bn;
    const int bno = p.geglu ? bn / 2 : bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bno - 1) / bno;
    const int tiles = MT * NT;
    if (const int mode = conv_gemm_bf16x_persistent_mode(p, cfg); mode >= 0) {
        const dim3 grid((unsigned)persistent_workgroups(), 1, 1);
        switch (cfg) {
            case 0: return launch_persist_bf16x<8, 5, 2, 4>(p, mode, grid, stream);
            case 1: return launch_persist_bf16x<8, 4, 2, 4>(p, mode, grid, stream);
            case 2: return launch_persist_bf16x<4, 4, 4, 2>(p, mode, grid, stream);
            case 3: return launch_persist_bf16x<4, 5, 2, 4>(p, mode, grid, stream);
        }
    }
    const dim3 grid = gemm_grid(p, tiles);
    switch (cfg) {
        case 0: return launch_cfg_bf16x<8, 5, 2, 4, -1>(p, grid, stream);
        case 1: return launch_cfg_bf16x<8, 4, 2, 4, -1>(p, grid, stream);
        case 2: return launch_cfg_bf16x<4, 4, 4, 2, -1>(p, grid, stream);
        case 3: return launch_cfg_bf16x<4, 5, 2, 4, -1>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
