// k_gemm3y.hip -- fp32 implicit-GEMM conv / linear on the bf16 matrix pipe, second kernel family: v_mfma_f32_32x32x16_bf16 on 32 x 160
// wave tiles.  EXPERIMENTAL (tiles 300 + x, chosen only by option gemm_tile): written at the end of round 2 from that round's last
// measurements, compiled and index-checked on the CPU (tests/test_gemm3y_layout_cpu.py), NOT yet run on an MI355X.
//
// Why.  k_gemm3x.hip keeps the matrix pipe 53-63 % busy, and round 2 ruled out everything outside the instruction stream: the k-tile head,
// hipcc's coarse LDS waits, DMA placement, wave priority, fabric traffic (L2 hit rate 85-91 % under the XCD cut at unchanged time) and the
// LDS-DMA path (a CU lands 57 B/clk from L2 and the kernel asks for 10-14) -- profiles/README.md, "Where the k loop's time goes".  What is
// left is the number of instructions between the matrix instructions: 430 per wave per k tile around 120 v_mfma_f32_16x16x32_bf16 (220 other
// VALU, of which 176 are the three-way split of the wave's 64 x 32 activation fragment), retiring at 8.4 cycles each.  This family halves both:
//   * v_mfma_f32_32x32x16_bf16: the same flops in half the matrix instructions (60 per wave per k tile), at the rate the data sheet quotes;
//   * wave tile 32 pixels x 160 channels (128 x 320 workgroup tile as 4 x 2 waves, 256 x 160 as 8 x 1): a wave splits 32 activation rows per
//     k tile instead of 64 (88 VALU instead of 176), and only 2 (or 1) instead of 4 waves split the same rows.
// Same arithmetic as k_gemm3x.hip (fp32 operands as exact sums of three bf16 terms, the six partial products >= 2^-24 of the product,
// fp32 accumulation; products in the order wl ah, wm am, wm ah, wh al, wh am, wh ah), same operands in HBM -- the activations as fp32 NHWC,
// the weights as the three bf16 planes launch_pack_split3 writes ([N][K / 32][plane][32], chunk g of a plane = k-tile elements
// {4g .. 4g+3, 16+4g .. 16+4g+3}) -- and the same epilogue semantics (bias + time-embedding row + residual, or a raw split-K slab).
//
// Operand layouts (v_mfma_f32_32x32x16_bf16 as k_attn_bf16.hip uses it; hi = lane >> 5, c = lane & 31):
//   A operand = weights: lane supplies output channel c of a 32-channel fragment, 8 k-values; B operand = activations: lane supplies pixel c
//   of the wave's 32, the same 8 k-values; D: lane holds pixel c and channels (r & 3) + 8 (r >> 2) + 4 hi of the fragment in acc[r], r < 16 --
//   four consecutive channels per r >> 2: 16-byte epilogue accesses.
//   Which 8 k-values: any assignment works as long as both operands use the same one.  A 32-deep k tile is two 16-deep matrix steps s;
//   lane half hi of step s takes plane chunk g = 2 s + hi = elements {4g .. 4g+3, 16+4g .. 16+4g+3} -- one ds_read_b128 per plane per
//   fragment from the planes as they are -- and, on the activation side, the two 16-byte fp32 chunks g and 4 + g of its pixel row.
// LDS: activation rows of 128 B, chunk c4 of tile row r in slot c4 ^ ((r >> 1) & 7) (the DMA applies the swizzle on the source address):
//   the 16-lane groups of a ds_read_b128 over 32 consecutive rows at one chunk ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) then cover
//   16 distinct 16-byte bank slots (even / odd rows in the two halves of the 256-byte bank window, (r >> 1) & 7 distinct within each).
//   Weight planes exactly as k_gemm3x.hip stages them: 16-column x 64-byte pieces, slot g ^ ((-(col >> 2)) & 3); a 32-column fragment is
//   two pieces, 1 KiB apart (bank-aligned), and the same argument makes its reads conflict-free.
// This first version keeps the plain loop structure (two LDS stages, one __syncthreads() per k tile, hipcc's own waits and schedule): round 2
// showed the structure of the loop is not what binds; the point to measure is the instruction count.
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

static const GemmTileInfo kTilesY[kNumGemmTilesY] = {{128, 320, "128x320y"}, {256, 160, "256x160y"}, {128, 256, "128x256y"}, {256, 128, "256x128y"}};
const GemmTileInfo& gemm_tile_info_y(int cfg) { return kTilesY[cfg]; }

template <int NI, int WM, int WN>
__global__ __launch_bounds__(512) void conv_gemm3y_kernel(const ConvGemm p) {
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * NI * WN;
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(BM % 64 == 0, "every wave issues whole 8-row DMA pieces of the activation tile");
    constexpr int NA = BM / 64;               // activation pieces (8 rows x 128 B) per wave per k tile
    constexpr int PW = (BN / 16) * 3;         // weight pieces (16 columns x 64 B: one plane of 16 channels) per k tile
    constexpr int NBW = (PW + 7) / 8;         // ... per wave
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE = A_BYTES + NBW * 8 * 1024;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_y[];

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
    const unsigned pix_bytes = (unsigned)p.a_ld * 4u;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Wbase = reinterpret_cast<const char*>(p.Bt3);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // ---- DMA sources (as k_gemm3x.hip; only the activation swizzle differs) ------------------------------------------------------------
    // activation piece j of a wave: tile rows (wave + 8 j) * 8 + sub, sub = lane >> 3; LDS slot lane & 7 receives global chunk
    // (lane & 7) ^ ((row >> 1) & 7), and (row >> 1) & 7 = ((wave & 1) * 4 + (sub >> 1)) & 7 whatever j is
    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ ((((wave & 1) << 2) + (sub >> 1)) & 7);
    int a_iy0[NA], a_ix0[NA];
    unsigned a_off[NA];
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
    // weight piece q = wave + 8 j: 16-column group q / 3, plane q % 3; lane -> column lane >> 2, LDS slot lane & 3 receives the plane's
    // 16-byte chunk (lane & 3) ^ ((-(column >> 2)) & 3)
    const unsigned w_row_bytes = (unsigned)p.kt_total * 192u;
    unsigned w_off[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int q = wave + 8 * j;
        const int f = q / 3, pl = q - 3 * f;
        const int r = lane >> 2;
        const int ch = (lane & 3) ^ ((-(r >> 2)) & 3);
        int n = n0 + f * 16 + r;
        // columns past N (ragged last tile) and the pieces past PW fetch the last valid row instead: real memory, never stored
        if (n >= p.N) n = p.N - 1;
        w_off[j] = (unsigned)n * w_row_bytes + pl * 64 + ch * 16;
    }
    int cs = kt_begin / T;
    const int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;

    // DMA of k tile kt_next into `stage`, straight-line (selects, no branches); then the source moves on to the next k tile -- unless
    // there is none: then the same tile is fetched once more, into the stage nobody reads any more (as k_gemm3x.hip's piece())
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
        for (int j = 0; j < NBW; ++j) {
            const char* src = Wbase + (w_off[j] + (unsigned)kt_next * 192u);
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

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------------
    const int c = lane & 31, hi = lane >> 5;
    const int arow = wm * 32 + c;                               // tile row of this lane's pixel
    const int asw = (arow >> 1) & 7;
    // activation chunks of matrix step s: g = 2 s + hi and 4 + g
    const int a_row_off = arow * 128;
    const int a_c0[2] = {a_row_off + (((0 + hi) ^ asw) << 4), a_row_off + (((2 + hi) ^ asw) << 4)};         // chunk g
    const int a_c1[2] = {a_row_off + (((4 + hi) ^ asw) << 4), a_row_off + (((6 + hi) ^ asw) << 4)};         // chunk 4 + g
    // weight fragment f (32 channels) of this wave: tile column wn * 32 NI + 32 f + c -> 16-column group, row inside it
    const int wcol0 = wn * 32 * NI + c;                         // + 32 f
    const int w_r16 = wcol0 & 15;                               // (32 f does not change it)
    const int w_sw = (-(w_r16 >> 2)) & 3;
    // byte offset inside the stage of plane pl, chunk g of fragment f: A_BYTES + ((wcol0 + 32 f) / 16 * 3 + pl) * 1024 + w_r16 * 64 + ((g ^ w_sw) << 4)
    const int w_base = A_BYTES + ((wcol0 >> 4) * 3) * 1024 + w_r16 * 64;     // fragment 0, plane 0; + f * 6144 + pl * 1024 + ((g ^ w_sw) << 4)
    const int w_g[2] = {((0 + hi) ^ w_sw) << 4, ((2 + hi) ^ w_sw) << 4};

    f32x16 acc[NI];
#pragma unroll
    for (int f = 0; f < NI; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    issue(smem_y);
    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        issue(smem_y + (cur ^ 1) * STAGE);
        const unsigned char* stage = smem_y + cur * STAGE;
        S3SplitT<true> sp[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(stage + a_c0[s]);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(stage + a_c1[s]);
            sp[s].load(x0, x1);
            sp[s].template steps<0, S3SplitT<true>::kSteps>();
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // planes in the order l, m, m, h, h, h: wl ah, wm am, wm ah, wh al, wh am, wh ah
            constexpr int WP3[6] = {2, 1, 1, 0, 0, 0};
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                const u32x4& a = (pr == 0 || pr == 2 || pr == 5) ? sp[s].h : ((pr == 3) ? sp[s].l : sp[s].m);
#pragma unroll
                for (int f = 0; f < NI; ++f) {
                    const u32x4 w = *reinterpret_cast<const u32x4*>(stage + w_base + f * 6144 + WP3[pr] * 1024 + w_g[s]);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc[f], 0, 0, 0);
                }
            }
        }
    }

    // the last k tile was fetched twice; that copy must have landed before the epilogue reuses the stages (and before the wave ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: bias + time-embedding row + residual (or a raw split-K slab), fp32 ------------------------------------------------------
    // Every wave transposes one 32 x 32 fragment at a time through its own LDS scratch (the stages are free now) and writes 128-byte row
    // segments with 16-byte lanes; the residual is read the same way.
    const bool split = p.splits > 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = !split && p.resid;
    const bool vec_ok = ((p.N & 3) == 0) && ((ldc & 3) == 0) && ((p.ldr & 3) == 0 || !has_resid);
    const int m_lane = m0 + arow;
    const int smp = (m_lane < p.M ? m_lane : 0) / HoWo;
    constexpr int LDSW = 36;                // scratch row stride in floats (32 + 4)
    if (vec_ok) {
        __syncthreads();                    // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_y + wave * (32 * LDSW * 4));
#pragma unroll
        for (int f = 0; f < NI; ++f) {
            const int nf0 = n0 + (wn * NI + f) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nf0 + 8 * q + 4 * hi;
                f32x4 v = {acc[f][4 * q], acc[f][4 * q + 1], acc[f][4 * q + 2], acc[f][4 * q + 3]};
                if (!split && n < p.N) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                }
                *reinterpret_cast<f32x4*>(scr + c * LDSW + 8 * q + 4 * hi) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), c4 = lane & 7;
                const int m = m0 + wm * 32 + row, n = nf0 + c4 * 4;
                if (m < p.M && n < p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                    if (has_resid) v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                    *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else if (m_lane < p.M) {
        // odd strides / N not a multiple of 4: element-wise stores straight from the accumulators
#pragma unroll
        for (int f = 0; f < NI; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * NI + f) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n < p.N) {
                    float sv = acc[f][r];
                    if (!split) {
                        if (p.bias) sv += p.bias[n];
                        if (p.rowvec) sv += p.rowvec[(long long)smp * p.rowvec_stride + n];
                        if (p.resid) sv += p.resid[(long long)m_lane * p.ldr + n];
                    }
                    Cf[(long long)m_lane * ldc + n] = sv;
                }
            }
    }
}

template <int NI, int WM, int WN>
static hipError_t launch_cfg_3y(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    static bool attr_set = false;
    auto k = conv_gemm3y_kernel<NI, WM, WN>;
    constexpr size_t lds = 2 * ((size_t)(32 * WM) * 128 + (size_t)(((32 * NI * WN / 16) * 3 + 7) / 8) * 8192);
    static_assert(lds <= 160 * 1024, "the stages must fit the CU's LDS");
    static_assert(8 * 32 * 36 * 4 <= lds, "epilogue scratch");
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm3y(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesY) return hipErrorInvalidValue;
    if ((p.Cin % 32) || p.CS != 32 || !p.zero_page || !p.Bt3 || p.out_mode != 0 || p.geglu || p.counters) return hipErrorInvalidValue;
    if ((unsigned long long)p.N * (unsigned long long)p.kt_total * 192ull >= 0xFFFFFF00ull) return hipErrorInvalidValue;   // 32-bit piece offsets
    const int bm = kTilesY[cfg].bm, bn = kTilesY[cfg].bn;
    const int tiles = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    const dim3 grid = gemm_grid(p, tiles);
    switch (cfg) {
        case 0: return launch_cfg_3y<5, 4, 2>(p, grid, stream);
        case 1: return launch_cfg_3y<5, 8, 1>(p, grid, stream);
        case 2: return launch_cfg_3y<4, 4, 2>(p, grid, stream);
        case 3: return launch_cfg_3y<4, 8, 1>(p, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
