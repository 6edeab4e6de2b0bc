// k_gemm_bf16t.hip -- the 256-row bf16 tiles of k_gemm_bf16x.hip for 3x3 / stride-1 / pad-1 convolutions, with the three taps of a kernel ROW served
// from ONE staged activation tile (precision = 1; bf16 tile_cfg 104 / 105).
//
// Why (profiles/r04c ... r04e, r04t): the bf16 k loop is bound by instruction issue -- an LDS-DMA instruction blocks its wave ~ 115 cycles -- and an
// implicit-GEMM 3x3 convolution stages the same pixels nine times, once per tap.  With 8 of 9 activation pieces dropped (an ablation) the UNet's 3x3
// shapes run +10 ... 12 %.  This kernel drops 2 of 3 without any change of data layout in HBM:
//   * the tile is 256 / W whole image rows.  The activation tile of kernel row ky is staged ONCE with each image row PADDED to W + 2 pixels (x = -1 ... W; the
//     two border columns, rows above / below the image and the tail of the last 8-row piece come from the zero page): staged row  yr (W + 2) + (x + 1).
//     Output pixel (yr, x) reads padded column x + kx at tap kx, so a 16-pixel fragment of tap kx is the 16 consecutive staged rows that start kx further
//     down -- the border is in the data, the k loop has no masks and no per-tap address work beyond one add;
//   * the XOR swizzle key of a staged row is its padded COLUMN (slot = chunk ^ (column & 7)), not its row index: a fragment starts at a column that is a
//     multiple of 16, so its key is (c + kx) & 7 for every fragment, and because every image row starts at an even staged row the (row parity, slot) pairs
//     of a ds_read_b128 are those of k_gemm_bf16x.hip's layout shifted by kx: conflict-free.  The source address carries the swizzle (the LDS side of the
//     DMA is lane-linear), one precomputed offset per piece.
// Per kernel row a wave issues 4-5 activation pieces instead of 12, i.e. (33 ... 36) + 3 x 40 = 153 ... 156 pieces per CU instead of 216 (-29 %).  LDS: two
// activation buffers of 264 ... 288 rows (one kernel row ahead, its pieces spread over the three taps of the current one) + two weight buffers (one tap ahead)
// = 149.5 ... 155.6 KB.  Everything else -- weight packing (k = (cs T + tap) 64 + ci), tile map, epilogue, the order of the products (results are bit-identical to tiles
// 100 / 101) -- is k_gemm_bf16x.hip's; split-K slices must hold whole kernel rows.
//
// (First form, measured in profiles/r04u: consecutive pixels staged unpadded and the border fixed by zeroing lane c = 0 / 15 of the fragments that start / end
// an image row -- 4 v_cndmask per such fragment: +2.5 % at W = 64, -3.5 % at W = 32, -8 % at W = 16: the masks cost more issue slots than the pieces saved.)
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

static const GemmTileInfo kTilesT[kNumGemmTilesT] = {{256, 320, "256x320t"}, {256, 256, "256x256t"}};
const GemmTileInfo& gemm_tile_info_t(int cfg) { return kTilesT[cfg]; }

// WF = W / 16: fragments per image row (1, 2, 4, 8)
template <int NI, int WF>
__global__ __launch_bounds__(512) void conv3_gemm_bf16t_kernel(const ConvGemm p) {
    constexpr int MI = 8, WM = 2, WN = 4;
    constexpr int BM = 16 * MI * WM;          // 256
    constexpr int BN = 16 * NI * WN;          // 320 / 256
    constexpr int W = 16 * WF;                // image width (launcher)
    constexpr int RPT = BM / W;               // image rows per tile
    constexpr int PADW = W + 2;               // staged pixels per image row
    constexpr int A_ROWS = (RPT * PADW + 7) / 8 * 8;   // 288 / 272 / 264 / 264
    constexpr int PA = A_ROWS / 8;            // 36 / 34 / 33 / 33 pieces per kernel row
    constexpr int NAJ = (PA + 7) / 8;         // <= 5 per wave (piece q = wave + 8 j)
    constexpr int NB = BN / 64;               // weight pieces per wave per tap
    constexpr int A_BYTES = A_ROWS * 128;
    constexpr int B_BYTES = BN * 128;
    static_assert(8 % WF == 0, "fragments of a wave tile start image rows at compile-time positions");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    unsigned char* const As = smem_t;                       // [2][A_ROWS][128]
    unsigned char* const Bs = smem_t + 2 * A_BYTES;         // [2][BN][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int MT = p.M / BM;                  // (launcher: M % BM == 0)
    const int NT = (p.N + BN - 1) / BN;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int m0 = gw.tm * BM;
    const int n0 = gw.tn * BN;
    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;                // (launcher: a multiple of 3)
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_g = (kt_end - kt_begin) / 3;                // kernel rows (groups of three taps) of this k slice
    const int HW = p.Hs * p.Ws;
    const int HoWo = HW;

    const unsigned pix_bytes = (unsigned)p.a_ld * 2u;
    const char* Abase = reinterpret_cast<const char*>(p.A);
    const char* Bbase = reinterpret_cast<const char*>(p.Bt);
    const char* zero = reinterpret_cast<const char*>(p.zero_page);

    // DMA pieces: 8 rows x 128 B; lane -> row sub = lane >> 3 of the piece, LDS slot lane & 7
    const int sub = lane >> 3;
    const int nb = m0 / HW;                                 // the tile lies inside one image (launcher: HW % BM == 0)
    const int m0_pix = m0 - nb * HW;
    const int y0 = m0_pix / W;
    const unsigned a_img = (unsigned)nb * (unsigned)HW * pix_bytes;
    // piece j of this wave holds staged rows i = (wave + 8 j) 8 + sub = image row yr = i / (W + 2) of the tile, padded column col = i % (W + 2): source pixel
    // (y0 + yr + ky - 1, col - 1), chunk (lane & 7) ^ (col & 7).  a_off[j]: its byte offset inside the image at ky = 1; a_y[j]: its image row (far outside for
    // the border columns and the tail rows, so that the range check of the row sends them to the zero page)
    int a_off[NAJ], a_y[NAJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int i = (wave + 8 * j) * 8 + sub;
        const int yr = i / PADW;
        const int col = i - yr * PADW;
        const bool inside = yr < RPT && col >= 1 && col <= W;
        a_y[j] = inside ? y0 + yr : -4;
        a_off[j] = ((y0 + yr) * W + col - 1) * (int)pix_bytes + (((lane & 7) ^ (col & 7)) << 4);
    }
    const int chunk = (lane & 7) ^ sub;                     // weight pieces: k_gemm_bf16x.hip's row swizzle
    // weight rows of piece j: n0 + (wave + 8 j) 8 + sub (rows beyond N read the zero page)
    const int b_n0 = n0 + wave * 8 + sub;
    const unsigned b_row_bytes = (unsigned)p.b_ld * 2u;
    const unsigned b_off0 = (unsigned)b_n0 * b_row_bytes + chunk * 16;

    // activation piece j of kernel row (cs, ky) -> buffer buf
    auto issue_a = [&](int j, int cs, int ky, int buf) {
        if ((wave + 8 * j) < PA) {                          // (wave-uniform: only wave 0 has a fifth piece)
            const bool ok = (unsigned)(a_y[j] + ky - 1) < (unsigned)p.Hs;
            const unsigned off = a_img + (unsigned)(a_off[j] + (ky - 1) * W * (int)pix_bytes) + (unsigned)cs * 128u;
            const char* src = (ok ? Abase : zero) + (ok ? off : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(As + buf * A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int kt, int buf) {                   // the weight tile of k tile kt (one tap of one 64-channel slice)
        const unsigned k0b = (unsigned)kt * 128u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool ok = b_n0 + 64 * j < p.N;
            const char* src = (ok ? Bbase : zero) + (ok ? b_off0 + (unsigned)(64 * j) * b_row_bytes + k0b : 0u);
            __builtin_amdgcn_global_load_lds((global_cvoid*)src, (lds_void*)(Bs + buf * B_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };

    const int c15 = lane & 15, g4 = lane >> 4;
    const int a_base = wm * (16 * MI + 2 * (MI / WF)) * 128;   // fragment f of the wave starts at staged row f 16 + 2 (f / WF) of the wave's block
    const int b_base = wn * 16 * NI * 128;
    const int fb_off0 = c15 * 128 + (((0 + g4) ^ (c15 & 7)) << 4);
    const int fb_off1 = c15 * 128 + (((4 + g4) ^ (c15 & 7)) << 4);

    f32x4 acc[MI][NI];

    // k tile t of this slice = tap kx = t % 3 of kernel row g = t / 3 = (cs, ky) with cs = kt / 9, ky = (kt % 9) / 3
    int kt = kt_begin;
    int cs = kt / 9;
    int ky = (kt - cs * 9) / 3;
    int kx = 0, g = 0;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) issue_a(j, cs, ky, 0);
    issue_b(kt, kt & 1);
    gemm_acc_init_bf16<MI, NI, WM, WN, 0, (NI == 5), (NI == 5 ? 0 : 4)>(p, acc, m0, n0, wave, lane, HoWo);   // zero, or the residual tile (ConvGemm::resid_acc), behind the first DMAs
    const int n_t = 3 * n_g;
    for (int t = 0; t < n_t; ++t) {
        sdmi_dma_landed();        // (k_common.hpp: this wave's LDS-DMA pieces have landed BEFORE it enters the barrier)
        __syncthreads();                                     // everything issued so far has landed (k tile kt, this kernel row's activations); the buffers written next are free
        const int abuf = g & 1;
        const bool more_g = g + 1 < n_g;
        auto issue_next = [&]() {
            if (t + 1 < n_t) issue_b(kt + 1, (kt + 1) & 1);
            if (more_g) {                                        // the next kernel row's activations, spread over this row's three taps
                int cs1 = cs, ky1 = ky + 1;
                if (ky1 == 3) { ky1 = 0; ++cs1; }
                if (kx == 0) { issue_a(0, cs1, ky1, abuf ^ 1); issue_a(1, cs1, ky1, abuf ^ 1); }
                else if (kx == 1) { issue_a(2, cs1, ky1, abuf ^ 1); issue_a(3, cs1, ky1, abuf ^ 1); }
                else if (NAJ > 4) issue_a(4, cs1, ky1, abuf ^ 1);
            }
        };
        // Round 6 (ConvGemm::variant bit 2): the second wave of every SIMD (waves 4 - 7) issues its DMA pieces BETWEEN the tile's two k steps, not in front of them: behind the
        // barrier half of the waves read fragments and start the matrix pipe while the other half issue DMA -- every wave used to do both in the same order at the same
        // time (a 72 KB LDS read burst under an idle matrix pipe at the top of every k tile).  Same products in the same order: bit-identical.
        const bool late_dma = (p.variant & 4) && wave >= 4;
        if (!late_dma) issue_next();
        const unsigned char* sa = As + abuf * A_BYTES;
        const unsigned char* sb = Bs + (kt & 1) * B_BYTES;
        const int ck = c15 + kx;                             // fragment row c of tap kx: padded column (multiple of 16) + c + kx
        constexpr int GM = 4, NG = MI / GM;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1 && late_dma) issue_next();
            const int fo = ck * 128 + ((((kk ? 4 : 0) + g4) ^ (ck & 7)) << 4);
            const int fbo = kk ? fb_off1 : fb_off0;
            u32x4 fb[NI];
            u32x4 fa[2][GM];
            auto read_a = [&](int f) { return *reinterpret_cast<const u32x4*>(sa + a_base + (f * 16 + 2 * (f / WF)) * 128 + fo); };
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const u32x4*>(sb + b_base + ni * 2048 + fbo);
#pragma unroll
            for (int i = 0; i < GM; ++i) fa[0][i] = read_a(i);
#pragma unroll
            for (int gg = 0; gg < NG; ++gg) {
                if (gg + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < GM; ++i) fa[(gg + 1) & 1][i] = read_a((gg + 1) * GM + i);
                }
#pragma unroll
                for (int i = 0; i < GM; ++i)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[gg * GM + i][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ni]), __builtin_bit_cast(bf16x8, fa[gg & 1][i]),
                                                                                   acc[gg * GM + i][ni], 0, 0, 0);
            }
            // the issue order of k_gemm_bf16x.hip, pinned: the weight fragments and the first row group, then one read of the next group per row of matrix instructions
            __builtin_amdgcn_sched_group_barrier(0x100, NI + GM, 0);
#pragma unroll
            for (int gg = 0; gg < NG; ++gg)
#pragma unroll
                for (int i = 0; i < GM; ++i) {
                    if (gg + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, NI, 0);
                }
        }
        ++kt;
        if (++kx == 3) {
            kx = 0; ++g;
            if (++ky == 3) { ky = 0; ++cs; }
        }
    }

    gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_t, m0, n0, z, wave, lane, HoWo);
}

template <int NI, int WF>
static hipError_t launch_cfg_bf16t(const ConvGemm& p, dim3 grid, hipStream_t stream) {
    auto k = conv3_gemm_bf16t_kernel<NI, WF>;
    constexpr size_t lds = 2 * (size_t)(((256 / (16 * WF)) * (16 * WF + 2) + 7) / 8 * 8 + 64 * NI) * 128;
    static_assert(lds <= 160 * 1024, "two activation + two weight buffers must fit the CU's LDS");
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

// what the kernel takes: a 3x3 / stride-1 / pad-1 convolution without upsampling over images whose width is 16, 32, 64 or 128 and whose pixel count is a
// multiple of the 256-row tile (so a tile lies inside one image), bf16 storage, k slices of whole kernel rows, no GEGLU pairing
bool conv_gemm_bf16t_supported(const ConvGemm& p) {
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.ups != 0 || p.geglu) return false;
    if ((p.Cin % 64) || !p.zero_page) return false;
    if (p.Ws != 16 && p.Ws != 32 && p.Ws != 64 && p.Ws != 128) return false;
    if (p.Ho != p.Hs || p.Wo != p.Ws || (p.Hs * p.Ws) % 256 || p.M % 256) return false;
    if (p.kt_per_split % 3) return false;
    return true;
}

hipError_t launch_conv_gemm_bf16t(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTilesT || !conv_gemm_bf16t_supported(p)) return hipErrorInvalidValue;
    const int bm = kTilesT[cfg].bm, bn = kTilesT[cfg].bn;
    const int MT = p.M / bm, NT = (p.N + bn - 1) / bn;
    const dim3 grid = gemm_grid(p, MT * NT);
    const int wf = p.Ws / 16;
#define SDMI_T(NI_)                                                              \
    switch (wf) {                                                                \
        case 1: return launch_cfg_bf16t<NI_, 1>(p, grid, stream);                \
        case 2: return launch_cfg_bf16t<NI_, 2>(p, grid, stream);                \
        case 4: return launch_cfg_bf16t<NI_, 4>(p, grid, stream);                \
        case 8: return launch_cfg_bf16t<NI_, 8>(p, grid, stream);                \
    }
    if (cfg == 0) { SDMI_T(5) } else { SDMI_T(4) }
#undef SDMI_T
    return hipErrorInvalidValue;
}

}  // namespace sdmi
