// Flash-style attention on the bf16 matrix cores (precision = 1): qkv_attention of the UNet's
// self/cross attention (attention.rs:5-45, head dims 40 / 80 / 160) with bf16 q/k/v/o in HBM,
// fp32 scores, fp32 online softmax and fp32 accumulation.
//
// One wave owns 32 query rows and all of the head's d columns; a workgroup of NW waves shares
// 64-key K/V tiles through double-buffered LDS.
//
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16, A = K fragment (m = key), B = Q fragment (n = query):
//                 lane l then holds 16 of the 32 scores of query (l & 31) in a key tile --
//                 keys crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi, hi = l >> 5 -- so the softmax
//                 is lane-local except for ONE v_permlane32_swap per tile (row max).
//   O^T = V^T P^T the eight probabilities r = 8 s .. 8 s + 7 of a lane are, packed to bf16,
//                 exactly the B operand (k = 8 hi + j) of a 16-key step when the contraction
//                 index is read as key(s, hi, j) = 16 s + 8 (j >> 2) + 4 hi + (j & 3).  The A
//                 operand V^T[d][key(s, hi, j)] is two ds_read_b64_tr_b16 (the LDS transpose
//                 read): V stays row-major [key][d] in LDS, the way it arrives from HBM, and
//                 there is no P round trip and no cross-lane exchange at all.
//
// LDS images (bf16): K rows padded to an ODD number of 16-byte chunks, so the ds_read_b128 of
// 16 different rows at one chunk column fall in 16 different 16-byte bank slots; V rows at a
// stride = 64 (mod 256) bytes with 64-byte row segments, so the 4 rows x 64 B a half-wave
// transpose-read touches tile the 256-byte bank row exactly.
// Softmax work per score is ONE v_exp_f32, half a v_max3 and half a v_cvt_pk (round 4; the matrix pipe was 24 % busy at d = 40 because
// the softmax's VALU work was of the order of the matrix work):
//   * q arrives PRE-SCALED by d^-0.5 log2(e) (the scale of attention.rs:15-26 -- q and k each carry d^-0.25 there -- times log2 e): the engine
//     folds it into the query projection's weight at load, before that weight's only bf16 rounding, so the scores leave the matrix pipe in
//     log2 units and the kernel never multiplies them;
//   * the running row maximum m enters as the ACCUMULATOR INPUT of the first K Q^T instruction (a 16-register vector holding -m), so the
//     pipe returns s - m and the probabilities are exp2 of that, no subtraction;
//   * m is only raised when a tile's maximum exceeds it by more than 2^8 (deferred rescale: probabilities up to 256 are as exact in
//     bf16 / fp32 as those below 1): the rescale of O, m and the scores is a wave-uniform slow path;
//   * the row sum is a column of ones appended to V (d = 40 / 80: the 32-row tiles of O^T have spare rows), i.e. it is accumulated by the
//     P V instructions themselves, from the same bf16-rounded probabilities as the numerator.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cmath>

namespace sdmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short h16;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int D, int NW, int QR = 1>
struct AttnBfCfg {
    static constexpr int NT = NW * 64;
    static constexpr int BKV = (D <= 40 && QR == 1) ? 128 : 64;   // keys per tile: d = 40 is softmax-bound, fewer / longer iterations (QR = 2: two score blocks per 32 keys, 64 keys fit the registers)
    static constexpr int KT = BKV / 32;                      // 32-key score tiles per K/V tile
    static constexpr int ST = BKV / 16;                      // 16-key steps of P V
    static constexpr int DK = (D + 15) / 16 * 16;            // contraction width of K Q^T (48 / 80 / 160)
    static constexpr int KS = DK / 16;
    static constexpr int NDT = (D + 31) / 32;                // 32-row tiles of O^T (2 / 3 / 5)
    static constexpr int KCH = DK / 8;                       // 16-byte chunks per K row holding data or zeros
    static constexpr int RSK = (KCH | 1) * 16;               // K row stride, odd chunk count (112 / 176 / 336 B)
    static constexpr int RSV = (NDT * 64 <= 192) ? 192 : 320;  // V row stride = 64 (mod 256) bytes
    static constexpr int K_BYTES = BKV * RSK;
    static constexpr int V_BYTES = BKV * RSV;
    static constexpr int CHUNKS = BKV * (D / 8);             // 16-byte chunks of one K (or V) tile in HBM
    static constexpr int NLD = (CHUNKS + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = 2 * (size_t)(K_BYTES + V_BYTES);
    static_assert(NDT * 64 <= RSV, "V rows must hold every d tile");
};

__device__ __forceinline__ float partner_max(float v) {  // max with lane l ^ 32
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float partner_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));  // v_cvt_pk_bf16_f32 (RNE)
}

// WPE (round 6): waves per SIMD the register budget is cut for -- 2 with NW = 4 lets TWO 4-wave workgroups share a CU (2 x 77.8 KB of LDS at d = 40), each with its own
// barrier, so the two waves of a SIMD are in different phases of their tiles (one in its matrix instructions while the other exponentiates / stages / reads fragments)
// QR (round 6): 32-row query blocks per wave.  QR = 2: a wave owns 64 query rows -- every K / V^T fragment it reads from LDS feeds two matrix instructions (half the
// fragment reads and half the K / V staging per score), and the two blocks' chains (K Q^T -> maximum -> exp -> V^T P^T) are independent, so a wave that waits on one
// block's exponentials or matrix results has the other block's instructions to issue (profiles/r04z: with one block per wave the parts of a tile ADD).
template <int D, int NW, int WPE = 1, int QR = 1>
__global__ __launch_bounds__(NW * 64, WPE) void attn_bf16_kernel(const AttnParams p) {
    using Cfg = AttnBfCfg<D, NW, QR>;
    constexpr int NT = Cfg::NT, BKV = Cfg::BKV, KS = Cfg::KS, NDT = Cfg::NDT, RSK = Cfg::RSK, RSV = Cfg::RSV, NLD = Cfg::NLD;
    constexpr int KT = Cfg::KT, ST = Cfg::ST;
    constexpr int CPR = D / 8;  // chunks per HBM row

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf[];
    unsigned char* Ks = smem_bf;
    unsigned char* Vs = smem_bf + 2 * Cfg::K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int c = lane & 31;

    const int b = blockIdx.y / p.n_head;
    const int hh = blockIdx.y - b * p.n_head;
    int qrow[QR];
    bool q_ok[QR];
#pragma unroll
    for (int g = 0; g < QR; ++g) {
        qrow[g] = blockIdx.x * (32 * NW * QR) + (wave * QR + g) * 32 + c;
        q_ok[g] = qrow[g] < p.nq;
    }

    const h16* Qh = reinterpret_cast<const h16*>(p.q) + (long long)b * p.q_bs + hh * D;
    const h16* Kh = reinterpret_cast<const h16*>(p.k) + (long long)b * p.k_bs + hh * D;
    const h16* Vh = reinterpret_cast<const h16*>(p.v) + (long long)b * p.v_bs + hh * D;
    h16* Oh = reinterpret_cast<h16*>(p.o) + (long long)b * p.o_bs + hh * D;

    const int nk = p.kv_len ? p.kv_len[b] : p.nk;
    const int n_tiles = (nk + BKV - 1) / BKV;
    const int n_full = nk / BKV;
    constexpr bool SUMCOL = (NDT * 32 > D);       // a spare row of O^T holds the row sum (V gets a column of ones)
    constexpr int SUM_DT = D / 32, SUM_R = ((D % 32) >> 3) * 4;   // ... row D: tile D / 32, register 4 (D % 32 / 8) of the lanes with hi = 0
    static_assert(!SUMCOL || (D % 8 == 0 && (D % 32) % 8 == 0), "row D of O^T must sit in a lane with hi = 0");
    constexpr float kDefer = 8.0f;                // raise the running maximum only past 2^8

    if constexpr (SUMCOL) {   // V[key][D] = 1, V[key][D + 1 .. D + 7] = 0 in both buffers, once (the staging never touches that chunk)
        for (int i = tid; i < 2 * BKV; i += NT)
            *reinterpret_cast<u32x4*>(Vs + (i / BKV) * Cfg::V_BYTES + (i % BKV) * RSV + CPR * 16) = u32x4{0x3F80u, 0u, 0u, 0u};
    }

    if constexpr (Cfg::DK > D) {  // zero the K columns D..DK-1 once (the staging never touches them)
        for (int i = tid; i < 2 * BKV; i += NT)
            *reinterpret_cast<u32x4*>(Ks + (i / BKV) * Cfg::K_BYTES + (i % BKV) * RSK + CPR * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    u32x4 rk[NLD], rv[NLD];
    auto gload = [&](int tile) {
        const int kv0 = tile * BKV;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR;
            const int c8 = idx - row * CPR;
            const int key = kv0 + row;
            u32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (idx < Cfg::CHUNKS && key < nk) {
                kk = *reinterpret_cast<const u32x4*>(Kh + (long long)key * p.ldk + c8 * 8);
                vv = *reinterpret_cast<const u32x4*>(Vh + (long long)key * p.ldv + c8 * 8);
            }
            rk[i] = kk;
            rv[i] = vv;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            if (idx < Cfg::CHUNKS) {
                const int row = idx / CPR;
                const int c8 = idx - row * CPR;
                *reinterpret_cast<u32x4*>(Ks + buf * Cfg::K_BYTES + row * RSK + c8 * 16) = rk[i];
                *reinterpret_cast<u32x4*>(Vs + buf * Cfg::V_BYTES + row * RSV + c8 * 16) = rv[i];
            }
        }
    };

    f32x16 o[QR][NDT];
    f32x16 negm[QR];          // -m in every register: the accumulator input of K Q^T (m = the row's reference maximum, log2 units; 0 before the first tile)
    float l_run[QR];          // d = 160 (no spare row): this lane's share of the row sum
#pragma unroll
    for (int g = 0; g < QR; ++g) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][dt][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[g][r] = 0.f;
        l_run[g] = 0.f;
    }

    // per-lane LDS offsets: K fragment row c, chunk hi; V transpose-read row 4 hi + (i >> 2), columns 16 (G & 1) + 4 (i & 3)
    const int k_off = c * RSK + hi * 16;
    const int i16 = lane & 15;
    const int v_off = (4 * hi + (i16 >> 2)) * RSV + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;

    gload(0);     // (round 6: the first K / V tile is requested before the query rows, k_attn_split.hip)
    // Q^T fragments: B[k = 16 s + 8 hi + j][n = query]
    bf16x8 qf[QR][KS];
#pragma unroll
    for (int g = 0; g < QR; ++g)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const int col = 16 * s + 8 * hi;
            if (q_ok[g] && col < D) v = *reinterpret_cast<const u32x4*>(Qh + (long long)qrow[g] * p.ldq + col);
            qf[g][s] = __builtin_bit_cast(bf16x8, v);
        }

    lstore(0);
    __syncthreads();

    for (int tile = 0; tile < n_tiles; ++tile) {
        const int cur = tile & 1;
        const bool more = (tile + 1) < n_tiles;

        const unsigned char* Kt = Ks + cur * Cfg::K_BYTES + k_off;
        const unsigned char* Vt = Vs + cur * Cfg::V_BYTES + v_off;
        const int kv0 = tile * BKV;

        // S^T = K Q^T.  The K fragments of 32-key tile kt + 1 are read before the MFMAs of tile kt are issued (fenced: hipcc
        // otherwise reads each fragment right in front of its MFMA and the wave sits out one LDS latency per MFMA).
        // (d = 160: ten fragments per tile -- a second set does not fit the register file; read in place as before)
        constexpr bool PF = KS <= 5;
        f32x16 s[QR][KT];
        bf16x8 kf[PF ? 2 : 1][KS];
        if constexpr (PF) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[0][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Kt + ks * 32));
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if constexpr (PF) {
                if (kt + 1 < KT) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        kf[(kt + 1) & 1][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Kt + (kt + 1) * 32 * RSK + ks * 32));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[0][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Kt + kt * 32 * RSK + ks * 32));
            }
#pragma unroll
            for (int g = 0; g < QR; ++g) s[g][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PF ? (kt & 1) : 0][0], qf[g][0], negm[g], 0, 0, 0);     // s - m
#pragma unroll
            for (int ks = 1; ks < KS; ++ks)
#pragma unroll
                for (int g = 0; g < QR; ++g) s[g][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PF ? (kt & 1) : 0][ks], qf[g][ks], s[g][kt], 0, 0, 0);
            if constexpr (PF) __builtin_amdgcn_sched_barrier(0);
        }

        // (round 6, measured and not kept -- profiles/r06zb_*: the second wave of every SIMD one phase late (its V^T P^T of tile t - 1 at the top of iteration t, a third V buffer, the
    // packed probabilities carried across the barrier; bit-identical, 74 operator cases): 481 -> 522 us, attention class 7.32 -> 7.80 ms per bf16 image -- what pays in the GEMMs'
    // k loops (staggered DMA issue) does not pay here; profiles/r06s_*: 128-key LDS tiles walked as two 64-key compute chunks by the two-block form, one barrier and two staging rounds per
    // 128 keys: attention class 6.80 -> 7.09 ms per bf16 image; profiles/r06k_*: requesting tile t + 2's rows during tile t through a second staging register set changes nothing, 490.7 vs 488.6 us:
    // the staging loads are not what a tile waits for)
    // the next tile's global loads go out here, not at the top of the iteration: hipcc puts a vmcnt(0) in front of the first
        // MFMA of the loop body, which made every tile wait for the loads it had just issued; they are consumed by lstore() below
        if (more) gload(tile + 1);

        if (tile >= n_full) {  // ragged last tile (uniform)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= nk) {
#pragma unroll
                        for (int g = 0; g < QR; ++g) s[g][kt][r] = -INFINITY;
                    }
                }
        }

        float mt[QR], alpha[QR];
        bool moves = tile == 0;
#pragma unroll
        for (int g = 0; g < QR; ++g) {
            float m1 = s[g][0][0];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m1 = fmaxf(m1, s[g][kt][r]);
            mt[g] = partner_max(m1);         // the tile's maximum relative to m
            alpha[g] = 1.0f;
            moves = moves || mt[g] > kDefer;
        }
        if (tile == 0 || __any(moves)) {   // wave-uniform: a reference maximum moves (always on the first tile, where it is still 0)
#pragma unroll
            for (int g = 0; g < QR; ++g) {
                const float delta = tile == 0 ? mt[g] : fmaxf(mt[g], 0.f);
                if (tile != 0) {
                    alpha[g] = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) o[g][dt] *= alpha[g];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[g][r] -= delta;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[g][kt][r] -= delta;
            }
        }

        unsigned pb[QR][ST][4];  // [16-key step][4 dwords = 8 bf16]
        float psum[QR];
#pragma unroll
        for (int g = 0; g < QR; ++g) psum[g] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#pragma unroll
                for (int g = 0; g < QR; ++g) {     // the blocks' exponentials side by side: independent instructions next to each other
                    const float e0 = __builtin_amdgcn_exp2f(s[g][kt][r]);
                    const float e1 = __builtin_amdgcn_exp2f(s[g][kt][r + 1]);
                    if constexpr (!SUMCOL) psum[g] += e0 + e1;
                    pb[g][kt * 2 + (r >> 3)][(r & 7) >> 1] = pack_bf16(e0, e1);
                }
            }
        }
        if constexpr (!SUMCOL) {
#pragma unroll
            for (int g = 0; g < QR; ++g) l_run[g] = l_run[g] * alpha[g] + psum[g];
        }

        // O^T += V^T P^T, 16 keys at a time; the V^T fragments of step st + 1 are read before the MFMAs of step st (fenced, as above)
        auto read_vf = [&](bf16x8 (&vf)[NDT], int st) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const unsigned char* vb = Vt + st * 16 * RSV + dt * 64;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>((const lds_s16x4*)vb));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>((const lds_s16x4*)(vb + 8 * RSV)));
                vf[dt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        };
        bf16x8 vf[PF ? 2 : 1][NDT];
        if constexpr (PF) read_vf(vf[0], 0);
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            if constexpr (PF) {
                if (st + 1 < ST) read_vf(vf[(st + 1) & 1], st + 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                read_vf(vf[0], st);
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int g = 0; g < QR; ++g) {
                    const u32x4 pw = {pb[g][st][0], pb[g][st][1], pb[g][st][2], pb[g][st][3]};
                    o[g][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[PF ? (st & 1) : 0][dt], __builtin_bit_cast(bf16x8, pw), o[g][dt], 0, 0, 0);
                }
            if constexpr (PF) __builtin_amdgcn_sched_barrier(0);
        }

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int g = 0; g < QR; ++g) {
        float l_own = l_run[g];
        if constexpr (SUMCOL) l_own = hi ? 0.f : o[g][SUM_DT][SUM_R];     // row D of O^T = sum of the probabilities
        const float inv = 1.0f / partner_sum(l_own);
        if (q_ok[g]) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int dcol = 32 * dt + 8 * rq + 4 * hi;
                    if (dcol < D) {
                        const u32x2 w = {pack_bf16(o[g][dt][4 * rq] * inv, o[g][dt][4 * rq + 1] * inv),
                                         pack_bf16(o[g][dt][4 * rq + 2] * inv, o[g][dt][4 * rq + 3] * inv)};
                        *reinterpret_cast<u32x2*>(Oh + (long long)qrow[g] * p.ldo + dcol) = w;
                    }
                }
            }
        }
    }
}

template <int D, int NW, int WPE = 1, int QR = 1>
static hipError_t launch_attn_bf16_d(const AttnParams& p, hipStream_t stream) {
    auto k = attn_bf16_kernel<D, NW, WPE, QR>;
    const size_t lds = AttnBfCfg<D, NW, QR>::LDS_BYTES;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    dim3 grid((p.nq + 32 * NW * QR - 1) / (32 * NW * QR), p.n * p.n_head);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_attn_bf16_any(const AttnParams& p, hipStream_t stream) {
    // widest workgroup that still gives every CU a workgroup (256 CUs)
    const long long bh = (long long)p.n * p.n_head;
    // Round 6 forms (AttnParams::variant; 0x100 = tests: the form whatever the grid size).  Measured at CFG batch 16 (profiles/r06d_*, r06e_*):
    //   bit 1  d = 40, self attention: 64 query rows per wave on 8-wave workgroups -- 64 x 64: 555 -> 476 us (619 -> 721 TFLOP/s);
    //   bit 2  d = 40, a context of at most two 64-key tiles: the same on 4-wave workgroups, two per CU -- 4096 x 77: 50.5 -> 34.6 us;
    //   bit 0  d = 80 (and d = 40 without bits 1 / 2): 4-wave workgroups, two per CU -- 1024 x 77: 23.2 -> 19.7 us; no gain on the self attentions, so short contexts only.
    const bool force = (p.variant & 0x100) != 0;
    const bool short_ctx = p.nk <= 128;
    if constexpr (D == 40) {
        if ((p.variant & 4) && (force ? !(p.variant & 2) : (short_ctx && (long long)((p.nq + 255) / 256) * bh >= 512))) return launch_attn_bf16_d<D, 4, 2, 2>(p, stream);
        if ((p.variant & 2) && (force || (!short_ctx && (long long)((p.nq + 511) / 512) * bh >= 256))) return launch_attn_bf16_d<D, 8, 2, 2>(p, stream);
    }
    if constexpr (D == 40 || D == 80) {
        if ((p.variant & 1) && (force || (short_ctx && (long long)((p.nq + 127) / 128) * bh >= 512))) return launch_attn_bf16_d<D, 4, 2>(p, stream);
    }
    if ((long long)((p.nq + 255) / 256) * bh >= 256) return launch_attn_bf16_d<D, 8>(p, stream);
    if ((long long)((p.nq + 127) / 128) * bh >= 256) return launch_attn_bf16_d<D, 4>(p, stream);
    return launch_attn_bf16_d<D, 2>(p, stream);
}

// bf16 matrix-core attention; p.bf16 must be set, no additive mask (the masked CLIP path is fp32).  q must arrive multiplied by
// d_head^-0.5 log2(e) (kernel header); p.scale is not used.
hipError_t launch_attention_bf16(const AttnParams& p, hipStream_t stream) {
    if (!p.bf16 || p.mask) return hipErrorInvalidValue;
    if (!p.q_log2) return hipErrorInvalidValue;   // the kernel applies no scale: a q without attn_bf16_q_scale folded in would give a silently wrong softmax
    if ((p.ldq | p.ldk | p.ldv | p.ldo) & 7) return hipErrorInvalidValue;  // 16-byte row alignment
    switch (p.d_head) {
        case 40: return launch_attn_bf16_any<40>(p, stream);
        case 80: return launch_attn_bf16_any<80>(p, stream);
        case 160: return launch_attn_bf16_any<160>(p, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
