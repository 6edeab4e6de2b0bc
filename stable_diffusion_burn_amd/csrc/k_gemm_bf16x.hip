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

// ---- PIPE = 2 (option gemm_bf16x_variant = 3): the k loop as one gap-free matrix-instruction stream ------------------------------
// The loop below the barrier of the plain form opens every k tile with the DMA of the next tile in one block (address arithmetic
// with exec-mask branches for the padding taps: ~150 instructions for 9 pieces) and the reads of the first fragments -- more than
// a thousand cycles in which neither wave of a SIMD issues a matrix instruction, against 2 560 cycles of matrix work per tile.
// Here a k tile is a straight line of 2 MI fragment rows (two 32-deep k steps x MI rows) of NI matrix instructions, and everything
// else sits in the issue slots behind them, pinned with sched_barrier(0) as in k_gemm3x.hip:
//   * the next tile's DMA pieces, branch-free (zero page by select, 32-bit offsets), one every other slot of the first rows;
//   * activation fragments through a ring of R registers: the fragment of row r + R is read behind the last instruction of row r;
//   * weight fragments WITHOUT a second buffer: in the last row of a k step fragment ni is dead as soon as its matrix instruction
//     has issued, and the next k step's fragment ni is read into it right there, NI instructions ahead of its use;
//   * the barrier "the next tile has landed" therefore moves from the top of the tile to the end of row 2 MI - R, the last point
//     before a read of the next tile; by then this wave has issued -- and waited for -- every read of the current stage, so the barrier
//     still doubles as "this stage may be overwritten", and all DMA pieces were issued rows ago: vmcnt(0) finds them landed.
// The waits: hipcc's own LDS waits in a kernel with LDS-DMA in flight are all `s_waitcnt lgkmcnt(0)`, so the wait in front of the first use of
// a fragment also waits for the fragment read that was issued one slot ago -- an LDS round trip at every row of the rolling schedule above
// (a first form of this loop used hipcc's waits: bit-identical, not faster, removed; profiles/r02y_ab_bf16_b8_pipelined_loop.jsonl).  So the
// fragment reads are inline asm (invisible to the waitcnt pass) and the waits are
// COUNTED by hand: LDS reads of a wave complete in issue order, so `lgkmcnt(n)` in front of a matrix instruction is exactly "everything but
// the n reads issued after my operands has landed".  n is computed at compile time from the schedule itself (reads_in / wait_n below);
// tools/dev/check_lgkm.py re-derives the guarantee from the COMPILED instruction stream (every use of a fragment register is behind a
// wait that covers its read), and the parity tests hold the loop bit-identical to the plain one.
template <int MI, int NI, int NA, int NB, int A_BYTES>
struct BxWave {
    static constexpr int ROWS = 2 * MI;
    static constexpr int R = (MI >= 8) ? 4 : 2;
    static constexpr int BAR_ROW = ROWS - R;
    static constexpr int NP = NA + NB;
    static constexpr int DMA_ROWS = (BAR_ROW < 4) ? BAR_ROW : 4;
    static_assert(R >= 2 && R <= MI && DMA_ROWS <= BAR_ROW && DMA_ROWS * NI >= NP, "ring / barrier / DMA placement");

    f32x4 acc[MI][NI];
    u32x4 fb[NI];
    u32x4 fa[R];
    int a_iy0[NA], a_ix0[NA];
    unsigned a_off[NA];               // byte offset of the sample + this lane's chunk (operands are < 4 GiB: launch_gemm checks)
    unsigned b_off[NB];
    const char *Abase, *Bbase, *zero;
    unsigned pix_bytes;
    int Hin, Win, ups, Ws, KH, KW, wave;
    int cs, ky, kx, kt_next, kt_end;
    const unsigned char *a_tile, *b_tile;     // current stage + this wave's activation / weight rows
    unsigned char* next_stage;                // where the DMA of k tile kt_next goes
    int fr_off0, fr_off1;

    // ---- the schedule's own arithmetic (slots are numbered r * NI + ni and continue across tiles, periodically) -------------
    static constexpr int TILE = ROWS * NI;
    static constexpr int fmod(int a, int m) { return ((a % m) + m) % m; }
    static constexpr bool roll_row(int r) { return fmod(r, MI) == MI - 1; }            // its slots re-load the weight fragments
    static constexpr int cnt(int s) { return (roll_row(fmod(s, TILE) / NI) ? 1 : 0) + (fmod(s, NI) == NI - 1 ? 1 : 0); }   // reads issued behind slot s's matrix instruction
    static constexpr int reads_in(int a, int b) { int n = 0; for (int u = a; u < b; ++u) n += cnt(u); return n; }
    // reads issued after the operands of matrix instruction (r, ni) and before it: the count its wait may leave outstanding
    static constexpr int newer_than_a(int r, int ni) { return reads_in((r - R) * NI + NI, r * NI + ni); }     // fa[r % R]: last read of slot (r - R, NI - 1)
    static constexpr int newer_than_b(int r, int ni) {                                                            // fb[ni]: first read of slot (rho, ni), rho = the roll row before r
        const int rho = (r >= MI) ? MI - 1 : -1;
        return (ni == NI - 1 ? 1 : 0) + reads_in(rho * NI + ni + 1, r * NI + ni);
    }
    static constexpr int wait_n(int r, int ni) {     // -1: an earlier wait of the row covers this instruction's operands
        const int a = newer_than_a(r, ni), b = newer_than_b(r, ni);
        int n = -1;
        if (ni == 0) n = a < b ? a : b;
        else if (r == 0 || r == MI) n = b;
        return n > 15 ? 15 : n;                      // lgkmcnt is a 4-bit field
    }
    unsigned a_lds[2], b_lds[2], an_lds, bn_lds;     // LDS byte addresses (+ this lane's chunk) of the wave's rows: current tile k step 0 / 1, next tile k step 0
    template <int OFF>
    static __device__ __forceinline__ void rd_asm(u32x4& dst, unsigned addr) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
    }

    // DMA piece J of k tile kt_next -> next_stage: straight-line code as in k_gemm3x.hip (S3Wave::piece); after the last piece the
    // source moves on to the next k tile, unless there is none: then the same tile is fetched once more into the stage nobody reads
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
            const char* src = Bbase + (b_off[j] + (unsigned)kt_next * 128u);
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

    static __device__ __forceinline__ u32x4 rd(const unsigned char* q) { return *reinterpret_cast<const u32x4*>(q); }

    // slot (row RW, column fragment NIX): one matrix instruction and what rides behind it
    template <int RW, int NIX>
    __device__ __forceinline__ void slots() {
        if constexpr (RW < ROWS) {
            constexpr int kk = RW / MI, mi = RW % MI;
            constexpr int WN_ = wait_n(RW, NIX);
            if constexpr (WN_ >= 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fb[NIX]), "+v"(fa[RW % R]) : "n"(WN_));
            acc[mi][NIX] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[NIX]), __builtin_bit_cast(bf16x8, fa[RW % R]),
                                                                  acc[mi][NIX], 0, 0, 0);
            if constexpr (RW < DMA_ROWS) {
                constexpr int sl = RW * NI + NIX, SL = DMA_ROWS * NI;
                pieces<sl * NP / SL, (sl + 1) * NP / SL>();
            }
            if constexpr (mi == MI - 1) {     // last row of a k step: weight fragment NIX of the next k step into the register just used
                if constexpr (kk == 0) rd_asm<NIX * 2048>(fb[NIX], b_lds[1]);
                else rd_asm<NIX * 2048>(fb[NIX], bn_lds);
            }
            if constexpr (NIX == NI - 1) {
                if constexpr (RW == BAR_ROW) {
                    // every read of the current stage has been issued (the last ones a row ago) and every DMA piece of the next tile rows ago
                    __builtin_amdgcn_sched_barrier(0);
                    // (the reads of the current stage were issued a row or more ago; waiting for them all here costs nothing and
                    // keeps "every wave is done with this stage" literally true at the barrier)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                constexpr int nr = RW + R;     // the activation fragment this ring slot holds next
                if constexpr (nr < ROWS) rd_asm<(nr % MI) * 2048>(fa[RW % R], a_lds[nr / MI]);
                else rd_asm<(nr - ROWS) * 2048>(fa[RW % R], an_lds);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NIX + 1 < NI) slots<RW, NIX + 1>();
            else slots<RW + 1, 0>();
        }
    }
    __device__ __forceinline__ void tile() {
        __builtin_amdgcn_sched_barrier(0);
        slots<0, 0>();
    }
    // the first k tile of the launch: what the tail of a tile does for its successor
    __device__ __forceinline__ void head() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = rd(b_tile + ni * 2048 + fr_off0);
#pragma unroll
        for (int i = 0; i < R; ++i) fa[i] = rd(a_tile + i * 2048 + fr_off0);
        {
            // these reads are hipcc's (it waits for them itself at their first use, which the asm waits below do not know): make them land
            // before the counted schedule starts, so that from here on the only LDS reads in flight are the loop's own
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(fb[ni]));
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(fa[i]));
        }
    }
    __device__ __forceinline__ void set_lds(unsigned cur_stage, unsigned nxt_stage, unsigned a_base, unsigned b_base) {   // LDS byte addresses of the stages
        a_lds[0] = cur_stage + a_base + (unsigned)fr_off0;
        a_lds[1] = cur_stage + a_base + (unsigned)fr_off1;
        b_lds[0] = cur_stage + b_base + (unsigned)fr_off0;
        b_lds[1] = cur_stage + b_base + (unsigned)fr_off1;
        an_lds = nxt_stage + a_base + (unsigned)fr_off0;
        bn_lds = nxt_stage + b_base + (unsigned)fr_off0;
    }
};

template <int MI, int NI, int WM, int WN, int PIPE>
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
    BxWave<MI, NI, NA, NB, BM * 128> w;       // the accumulators live here in both forms; the plain loop uses nothing else of it
    auto& acc = w.acc;
    const int c15 = lane & 15, g4 = lane >> 4;
    if constexpr (PIPE) {
        w.Hin = p.Hs << p.ups;
        w.Win = p.Ws << p.ups;
        w.ups = p.ups;
        w.Ws = p.Ws;
        w.KH = p.KH;
        w.KW = p.KW;
        w.wave = wave;
        w.pix_bytes = (unsigned)p.a_ld * 2u;
        w.Abase = reinterpret_cast<const char*>(p.A);
        w.Bbase = reinterpret_cast<const char*>(p.Bt);
        w.zero = reinterpret_cast<const char*>(p.zero_page);
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
            // rows past N (ragged last tile) fetch the last valid row instead: real memory, and the accumulator columns they feed are never stored
            if (n >= p.N) wrow -= (n - (p.N - 1));
            w.b_off[j] = (unsigned)wrow * (unsigned)p.b_ld * 2u + chunk * 16;
        }
        w.cs = kt_begin / T;
        const int tap0 = kt_begin - w.cs * T;
        w.ky = tap0 / p.KW;
        w.kx = tap0 - w.ky * p.KW;
        w.kt_next = kt_begin;
        w.kt_end = kt_end;
        w.fr_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
        w.fr_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);
        const int a_base = wm * 16 * MI * 128;
        const int b_base = BM * 128 + wn * 16 * NI * 128;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        w.next_stage = smem_x;
        w.template pieces<0, NA + NB>();      // k tile 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        w.a_tile = smem_x + a_base;
        w.b_tile = smem_x + b_base;
        const unsigned lds0 = (unsigned)(unsigned long long)(lds_void*)smem_x;     // LDS byte address of the stages
        w.head();
        for (int t = 0; t < n_t; ++t) {
            const int cur = t & 1;
            w.next_stage = smem_x + (cur ^ 1) * STAGE;
            w.a_tile = smem_x + cur * STAGE + a_base;
            w.b_tile = smem_x + cur * STAGE + b_base;
            w.set_lds(lds0 + cur * STAGE, lds0 + (cur ^ 1) * STAGE, a_base, b_base);
            w.tile();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads past the last tile (never used) have landed
        // the last k tile was fetched twice (piece()); that copy must have landed before the epilogue reuses the stages
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
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
    const SlabStore slab(Cf, split ? p.slab_stride : 0, split && p.counters && p.slab_wt);
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
                        if (split) slab.store((long long)m * ldc + n, v);
                        else *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
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
    if (split && p.counters) {
        if (splitk_arrive(p.counters, lid, p.splits, reinterpret_cast<unsigned*>(smem_x), p.slab_wt != 0)) splitk_reduce_tile<true>(p, m0, n0, BM, BN);
    }
}

template <int MI, int NI, int WM, int WN, int PIPE>
static hipError_t launch_cfg_bf16x(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16x_kernel<MI, NI, WM, WN, PIPE>;
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
    // p.variant (option gemm_bf16x_variant) = 3: the pipelined k loop (BxWave: fragment reads as inline asm with hand-counted waits)
    if ((p.variant & 3) == 3) {
        switch (cfg) {
            case 0: return launch_cfg_bf16x<8, 5, 2, 4, 2>(p, grid, stream);
            case 1: return launch_cfg_bf16x<8, 4, 2, 4, 2>(p, grid, stream);
            case 2: return launch_cfg_bf16x<4, 4, 4, 2, 2>(p, grid, stream);
            case 3: return launch_cfg_bf16x<4, 5, 2, 4, 2>(p, grid, stream);
        }
    }
    switch (cfg) {
        case 0: return launch_cfg_bf16x<8, 5, 2, 4, 0>(p, grid, stream);
        case 1: return launch_cfg_bf16x<8, 4, 2, 4, 0>(p, grid, stream);
        case 2: return launch_cfg_bf16x<4, 4, 4, 2, 0>(p, grid, stream);
        case 3: return launch_cfg_bf16x<4, 5, 2, 4, 0>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
