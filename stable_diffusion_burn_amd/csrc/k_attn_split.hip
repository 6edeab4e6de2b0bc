// k_attn_split.hip -- fp32 qkv_attention (attention.rs:5-45) on the bf16 matrix pipe, precision = 0, head dims 40 / 80.
//
// The same idea as k_gemm3x.hip: an fp32 number is exactly the sum of three bf16 numbers (x = h + m + l, round-to-nearest
// splits), a bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32 -- so S = Q K^T and O = P V are computed
// with fp32 q / k / v / probabilities as six bf16 MFMAs per block (the partial products >= 2^-24 of the product; per-product
// error <= 2^-23 worst case, 2^-28 on average).  Scores, running max, exponentials, row sums and the output stay fp32; q/k/v/o are fp32 in HBM.
// v_mfma_f32_16x16x4_f32 retires 256 flop/clk/CU, six v_mfma_f32_32x32x16_bf16 per fp32 block 683: the fp32 flash kernel
// (k_attn.hip, 87 TFLOP/s at d = 40) is bound by the former.
//
// Structure and operand layouts are those of k_attn_bf16.hip (a wave owns 32 query rows, S^T = K Q^T so the softmax is
// lane-local, the packed probabilities ARE the B operand of O^T = V^T P^T, V^T fragments by the LDS transpose read), with
//   * K / V tiles of 64 keys staged through registers: a thread splits the 8 floats it loaded and writes three 16-byte chunks,
//     one per plane -- the LDS images are [buffer][plane][key][row stride] bf16, same strides as the bf16 kernel;
//   * Q fragments split once per workgroup into registers; the probabilities split right after the exponentials, 32 keys at a time.
// d = 40, round 5 -- the packed tail.  Columns 32..39 used to cost a whole 16-deep k step of K Q^T (8 of 16 slots zero) and a whole 32-row tile of O^T (8 of 32 rows used),
// six instructions each.  Now (i) the tail k step pairs planes along k: K plane 0 holds [K_h | K_m] and K plane 1 [K_h | K_l] in slots 32..39 | 40..47, the query operand
// holds the SAME eight columns in both lane halves, so [K_h | K_l] x [Q_l | Q_h], [K_h | K_m] x [Q_m | Q_m] and [K_h | K_m] x [Q_h | Q_h] are the six products in three
// instructions; (ii) the tail tile of V^T stacks planes along its rows: V plane 0 holds [V_h | V_m | V_l | 0] in columns 32..63, so one instruction per probability plane
// yields the rows V_h P, V_m P, V_l P -- all nine products (three more than before, each < 2^-16 of the product) in three instructions, summed per lane after the key loop.
// 15 + 9 instead of 18 + 12 instructions per 32 keys (-20 %), 10 + 8 fragment reads instead of 18 + 12.
// Round 5 -- the softmax of k_attn_bf16.hip (template flag LG, the form that runs): q is multiplied by scale^2 log2(e) before its split (scores in log2 units), the reference
// maximum enters as the accumulator input of a score tile's first instruction (the pipe returns s - m: one v_exp_f32 per probability, no multiply-add), it is raised only when a
// tile exceeds it by 2^8 (wave-uniform slow path), and the row sum is a column of ones in V's high plane, accumulated by the P V instructions in a spare row of the last tile.
// Per pair of probabilities 12 vector instructions instead of 16; with the packed tail: attention class 40.3 -> 34.9 ms per batch-1 image (profiles/r05u_*, r05v_*), the 64 x 64
// self attention 290 -> 222 us = 193 TFLOP/s, matrix pipe ~ 61 % busy (2 waves x 66 instructions x 32 cycles per 64-key tile against ~ 6 900 cycles measured).
// Measured (rocprofv3 --pmc, 2 x 8 heads x 4096^2, d = 40; before the packed tail): matrix pipe 56 % busy, 138 TFLOP/s against 95 for k_attn.hip; 8.4 bf16
// flops are issued per fp32 flop (6 products x 1.4 for padding d = 40 to 48 in K Q^T and to 64 in V^T P^T).  A software-pipelined
// variant (S(t+1) issued between the exponentials of tile t, staging between the MFMAs of V^T P^T) measured 3 % SLOWER and was
// removed: the loop is bound by total issue slots of the two waves per SIMD, not by the order inside one wave.  Explicit, fenced
// prefetch of the K / V^T fragments one step ahead (what helps k_attn_bf16.hip by 6-14 %) measured 6 % slower here: hipcc's own
// schedule already overlaps these reads with the six-MFMA groups.
#include "kernels.hpp"
#include "k_split3.hpp"

#include <hip/hip_runtime.h>

#include <cmath>

namespace sdmi {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int D, int NW>
struct AttnSpCfg {
    static constexpr int NT = NW * 64;
    static constexpr int BKV = 64;                           // keys per tile
    static constexpr int KT = BKV / 32;                      // 32-key score tiles per K/V tile
    static constexpr int DK = (D + 15) / 16 * 16;            // contraction width of K Q^T (48 / 80)
    static constexpr int KS = DK / 16;
    static constexpr int NDT = (D + 31) / 32;                // 32-row tiles of O^T (2 / 3)
    static constexpr int KCH = DK / 8;                       // 16-byte bf16 chunks per K row holding data or zeros
    static constexpr int RSK = (KCH | 1) * 16;               // K row stride, odd chunk count (112 / 176 B)
    static constexpr int RSV = 192;                          // V row stride = 64 (mod 256) bytes; NDT * 64 <= 192
    static constexpr int K_BYTES = BKV * RSK;                // one plane of one buffer
    static constexpr int V_BYTES = BKV * RSV;
    static constexpr int CHUNKS = BKV * (D / 8);             // 8-element chunks of one K (or V) tile
    static constexpr int NLD = (CHUNKS + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = 2 * 3 * (size_t)(K_BYTES + V_BYTES);
    static_assert(NDT * 64 <= RSV, "V rows must hold every d tile");
    static_assert(D % 8 == 0, "8-element chunks");
};

__device__ __forceinline__ float sp_partner_max(float v) {  // max with lane l ^ 32
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float sp_partner_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ unsigned sp_pack(float lo, float hi) {   // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// (a, b) -> packed bf16 pairs h, m, l with a = h.lo + m.lo + l.lo and b = h.hi + m.hi + l.hi exactly
__device__ __forceinline__ void sp_split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = sp_pack(a, b);
    const f32x2 r = f32x2{a, b} - f32x2{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    m = sp_pack(r[0], r[1]);
    const f32x2 r2 = r - f32x2{__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    l = sp_pack(r2[0], r2[1]);
}
__device__ __forceinline__ void sp_split8(const f32x4 x0, const f32x4 x1, u32x4& h, u32x4& m, u32x4& l) {
    unsigned a[4], b[4], c[4];
    sp_split2(x0[0], x0[1], a[0], b[0], c[0]);
    sp_split2(x0[2], x0[3], a[1], b[1], c[1]);
    sp_split2(x1[0], x1[1], a[2], b[2], c[2]);
    sp_split2(x1[2], x1[3], a[3], b[3], c[3]);
    h = u32x4{a[0], a[1], a[2], a[3]};
    m = u32x4{b[0], b[1], b[2], b[3]};
    l = u32x4{c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ bf16x8 sp_bf(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int D, int NW, bool PK = false, bool LG = false>
__global__ __launch_bounds__(NW * 64) void attn_split_kernel(const AttnParams p) {
    using Cfg = AttnSpCfg<D, NW>;
    constexpr int NT = Cfg::NT, BKV = Cfg::BKV, KS = Cfg::KS, NDT = Cfg::NDT, RSK = Cfg::RSK, RSV = Cfg::RSV, NLD = Cfg::NLD;
    constexpr int KT = Cfg::KT;
    constexpr int CPR = D / 8;  // chunks per row
    constexpr float kLog2e = 1.4426950408889634f;
    // d = 40 (round 5): the head's last 8 columns would occupy a 16-deep k step of K Q^T and a 32-row tile of O^T on their own, six matrix instructions each for
    // 8 useful columns.  Both are PACKED instead (kernel header): 3 instructions per tail k step, 3 per tail tile -- 15 + 9 instead of 18 + 12 per 32 keys.
    constexpr bool PACK = PK;
    // LG (round 5): the softmax of k_attn_bf16.hip.  q is multiplied by scale^2 log2(e) before its split, so the scores leave the matrix pipe in log2 units; the row's
    // reference maximum enters as the ACCUMULATOR INPUT of each score tile's first instruction (the pipe returns s - m: the probabilities are one v_exp_f32 each);
    // m is raised only when a tile exceeds it by 2^8 (a wave-uniform slow path rescales O, m and the scores -- the three bf16 planes carry 24 bits of a probability
    // whatever its magnitude); the row sum is a COLUMN OF ONES appended to V's high plane, accumulated by the P V instructions in a spare row of the last O^T tile.
    constexpr int SUM_R = PACK ? 12 : 4 * ((D % 32) / 8);   // ... that row: register SUM_R of the lanes with hi = 0 (packed tile: row 24; otherwise row D % 32)
    constexpr float kDefer = 8.0f;
    static_assert(!PK || D == 40, "the packed tail is the d = 40 form");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
    unsigned char* Ks = smem_sp;                            // [buffer][plane][key][RSK]
    unsigned char* Vs = smem_sp + 6 * Cfg::K_BYTES;         // [buffer][plane][key][RSV]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int c = lane & 31;

    const int b = blockIdx.y / p.n_head;
    const int hh = blockIdx.y - b * p.n_head;
    const int qrow = blockIdx.x * (32 * NW) + wave * 32 + c;
    const bool q_ok = qrow < p.nq;

    const float* Qf = p.q + (long long)b * p.q_bs + hh * D;
    const float* Kf = p.k + (long long)b * p.k_bs + hh * D;
    const float* Vf = p.v + (long long)b * p.v_bs + hh * D;
    float* Of = p.o + (long long)b * p.o_bs + hh * D;

    const int nk = p.kv_len ? p.kv_len[b] : p.nk;
    const int n_tiles = (nk + BKV - 1) / BKV;
    const int n_full = nk / BKV;
    const float cs = p.scale * p.scale * kLog2e;  // scores -> log2 units
    // key slices (kv_splits > 1): this workgroup's tiles [t_begin, t_end)
    const int n_split = p.kv_splits > 1 ? p.kv_splits : 1;
    const int tps = (n_tiles + n_split - 1) / n_split;
    const int t_begin = min((int)blockIdx.z * tps, n_tiles), t_end = min(t_begin + tps, n_tiles);

    if constexpr (PACK) {         // V plane 0, columns 56..63 (rows 24..31 of the packed tail tile) = (1, 0, .., 0) in both buffers, once: row 24 sums the probabilities (LG)
        for (int i = tid; i < 2 * BKV; i += NT)
            *reinterpret_cast<u32x4*>(Vs + (i / BKV) * 3 * Cfg::V_BYTES + (i % BKV) * RSV + 7 * 16) = u32x4{LG ? 0x3F80u : 0u, 0u, 0u, 0u};
    } else if constexpr (LG) {    // V[key][D] = 1 in the high plane, 0 in the other two, both buffers, once (the staging never touches that chunk)
        for (int i = tid; i < 6 * BKV; i += NT)
            *reinterpret_cast<u32x4*>(Vs + (i / BKV) * Cfg::V_BYTES + (i % BKV) * RSV + CPR * 16) = u32x4{(i / BKV) % 3 == 0 ? 0x3F80u : 0u, 0u, 0u, 0u};
    }
    if constexpr (!PACK && Cfg::DK > D) {  // zero the K columns D..DK-1 of every plane once (the staging never touches them)
        for (int i = tid; i < 6 * BKV; i += NT)
            *reinterpret_cast<u32x4*>(Ks + (i / BKV) * Cfg::K_BYTES + (i % BKV) * RSK + CPR * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    f32x4 rk[NLD][2], rv[NLD][2];
    auto gload = [&](int tile) {
        const int kv0 = tile * BKV;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / CPR;
            const int c8 = idx - row * CPR;
            const int key = kv0 + row;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rk[i][0] = z; rk[i][1] = z; rv[i][0] = z; rv[i][1] = z;
            if (idx < Cfg::CHUNKS && key < nk) {
                const float* ks = Kf + (long long)key * p.ldk + c8 * 8;
                const float* vs = Vf + (long long)key * p.ldv + c8 * 8;
                rk[i][0] = *reinterpret_cast<const f32x4*>(ks);
                rk[i][1] = *reinterpret_cast<const f32x4*>(ks + 4);
                rv[i][0] = *reinterpret_cast<const f32x4*>(vs);
                rv[i][1] = *reinterpret_cast<const f32x4*>(vs + 4);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            if (idx < Cfg::CHUNKS) {
                const int row = idx / CPR;
                const int c8 = idx - row * CPR;
                u32x4 h, m, l;
                sp_split8(rk[i][0], rk[i][1], h, m, l);
                unsigned char* kd = Ks + buf * 3 * Cfg::K_BYTES + row * RSK + c8 * 16;
                unsigned char* vd = Vs + buf * 3 * Cfg::V_BYTES + row * RSV + c8 * 16;
                const bool tail = PACK && c8 == CPR - 1;       // columns 32..39: K plane 0 gets [h | m], K plane 1 [h | l]; V plane 0 gets [h | m | l]
                *reinterpret_cast<u32x4*>(kd) = h;
                *reinterpret_cast<u32x4*>(tail ? kd + 16 : kd + Cfg::K_BYTES) = m;
                *reinterpret_cast<u32x4*>(tail ? kd + Cfg::K_BYTES + 16 : kd + 2 * Cfg::K_BYTES) = l;
                if (tail) *reinterpret_cast<u32x4*>(kd + Cfg::K_BYTES) = h;
                sp_split8(rv[i][0], rv[i][1], h, m, l);
                *reinterpret_cast<u32x4*>(vd) = h;
                *reinterpret_cast<u32x4*>(tail ? vd + 16 : vd + Cfg::V_BYTES) = m;
                *reinterpret_cast<u32x4*>(tail ? vd + 32 : vd + 2 * Cfg::V_BYTES) = l;
            }
        }
    };

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY;  // running row max, raw score units
    float l_run = 0.f;        // this lane's share of the row sum
    f32x16 negm;              // LG: -m in every register (m = the row's reference maximum, log2 units; 0 before the first tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    // round 6: the 8-wave d = 80 form holds 256 registers; there -m is ONE register, copied into a score tile's accumulator input in front of its first instruction
    // (32 v_mov per 64-key tile beside ~ 700 vector instructions) instead of sixteen live ones -- the same values, and the two spilled registers are gone
    constexpr bool NEGM1 = LG && D == 80 && NW == 8;
    float negm1 = 0.f;

    // per-lane LDS offsets: K fragment row c, chunk hi; V transpose-read row 4 hi + (i >> 2), columns 16 (G & 1) + 4 (i & 3)
    const int k_off = c * RSK + hi * 16;
    const int i16 = lane & 15;
    const int v_off = (4 * hi + (i16 >> 2)) * RSV + (16 * ((lane >> 4) & 1) + 4 * (i16 & 3)) * 2;

    // round 6: the first K / V tile is requested BEFORE the query rows (whose split then runs while it is in flight): one memory round trip in front of the key loop
    // instead of two -- 860 attention launches per batch-1 image, the short ones 12 - 23 us
    if (t_begin < t_end) gload(t_begin);
    // Q^T fragments, split: B[k = 16 s + 8 hi + j][n = query]
    u32x4 qh[KS], qm[KS], ql[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
        const bool tail = PACK && s == KS - 1;                 // packed tail step: BOTH lane halves hold columns 32..39
        const int col = tail ? 16 * s : 16 * s + 8 * hi;
        if (q_ok && col < D) {
            x0 = *reinterpret_cast<const f32x4*>(Qf + (long long)qrow * p.ldq + col);
            x1 = *reinterpret_cast<const f32x4*>(Qf + (long long)qrow * p.ldq + col + 4);
        }
        if constexpr (LG) { x0 *= cs; x1 *= cs; }
        sp_split8(x0, x1, qh[s], qm[s], ql[s]);
        if (tail && hi) ql[s] = qh[s];                         // the third operand of the tail step is [Q_l | Q_h] against the K image [K_h | K_l]
    }

    if (t_begin < t_end) lstore(0);
    __syncthreads();

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        const bool more = (tile + 1) < t_end;
        if (more) gload(tile + 1);

        const unsigned char* Kt = Ks + cur * 3 * Cfg::K_BYTES + k_off;
        const unsigned char* Vt = Vs + cur * 3 * Cfg::V_BYTES + v_off;
        const int kv0 = tile * BKV;

        // ---- S^T = K Q^T: six partial products, smallest first; the two 32-key tiles alternate (independent accumulators)
        f32x16 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = LG ? (NEGM1 ? negm1 : negm[r]) : 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 kf[KT][3];
            if (PACK && ks == KS - 1) {   // packed tail step: [K_h | K_l] x [Q_l | Q_h], then [K_h | K_m] x Q_m and x Q_h: the same six products in three instructions
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) kf[kt][pl] = *reinterpret_cast<const u32x4*>(Kt + pl * Cfg::K_BYTES + kt * 32 * RSK + ks * 32);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][1]), sp_bf(ql[ks]), s[kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][0]), sp_bf(qm[ks]), s[kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][0]), sp_bf(qh[ks]), s[kt], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[kt][pl] = *reinterpret_cast<const u32x4*>(Kt + pl * Cfg::K_BYTES + kt * 32 * RSK + ks * 32);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][2]), sp_bf(qh[ks]), s[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][0]), sp_bf(ql[ks]), s[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][1]), sp_bf(qm[ks]), s[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][1]), sp_bf(qh[ks]), s[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][0]), sp_bf(qm[ks]), s[kt], 0, 0, 0);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(kf[kt][0]), sp_bf(qh[ks]), s[kt], 0, 0, 0);
        }

        if (tile >= n_full) {  // ragged last tile (uniform)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= nk) s[kt][r] = -INFINITY;
                }
        }

        float mt = s[0][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[kt][r]);
        mt = sp_partner_max(mt);
        float mc = 0.f, alpha = 1.0f;
        if constexpr (LG) {
            if (tile == t_begin || __any(mt > kDefer)) {   // wave-uniform: the reference maximum moves (always on the first tile, where it is still 0)
                const float delta = tile == t_begin ? mt : fmaxf(mt, 0.f);
                if (tile != t_begin) {
                    alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
                }
                if constexpr (NEGM1) negm1 -= delta;
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[r] -= delta;
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] -= delta;
            }
        } else {
            const float m_new = fmaxf(m_run, mt);
            mc = m_new * cs;
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
            m_run = m_new;
            if (__any(alpha != 1.0f)) {  // the running max moved somewhere in the wave
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
            }
        }

        // ---- per 32-key tile: probabilities (fp32), their three-way split, and O^T += V^T P^T for its two 16-key steps
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            unsigned pa[2][4], pb[2][4], pc[2][4];     // [16-key step of this tile][4 dwords = 8 bf16] x (h, m, l)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(LG ? s[kt][r] : __builtin_fmaf(s[kt][r], cs, -mc));
                const float e1 = __builtin_amdgcn_exp2f(LG ? s[kt][r + 1] : __builtin_fmaf(s[kt][r + 1], cs, -mc));
                if constexpr (!LG) psum += e0 + e1;
                sp_split2(e0, e1, pa[r >> 3][(r & 7) >> 1], pb[r >> 3][(r & 7) >> 1], pc[r >> 3][(r & 7) >> 1]);
            }
            u32x4 ph[2], pm[2], pl[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                ph[h2] = u32x4{pa[h2][0], pa[h2][1], pa[h2][2], pa[h2][3]};
                pm[h2] = u32x4{pb[h2][0], pb[h2][1], pb[h2][2], pb[h2][3]};
                pl[h2] = u32x4{pc[h2][0], pc[h2][1], pc[h2][2], pc[h2][3]};
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int st = kt * 2 + h2;
                u32x4 vf[NDT][3];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int pn = 0; pn < ((PACK && dt == NDT - 1) ? 1 : 3); ++pn) {
                        const unsigned char* vb = Vt + pn * Cfg::V_BYTES + st * 16 * RSV + dt * 64;
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>((const lds_s16x4*)vb));
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>((const lds_s16x4*)(vb + 8 * RSV)));
                        vf[dt][pn] = __builtin_bit_cast(u32x4, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                if constexpr (PACK) {   // tile 0: six products; the packed tail tile (rows 0..7 = V_h, 8..15 = V_m, 16..23 = V_l of columns 32..39): all nine products in three
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][2]), sp_bf(ph[h2]), o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[1][0]), sp_bf(pl[h2]), o[1], 0, 0, 0);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][0]), sp_bf(pl[h2]), o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[1][0]), sp_bf(pm[h2]), o[1], 0, 0, 0);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][1]), sp_bf(pm[h2]), o[0], 0, 0, 0);
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[1][0]), sp_bf(ph[h2]), o[1], 0, 0, 0);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][1]), sp_bf(ph[h2]), o[0], 0, 0, 0);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][0]), sp_bf(pm[h2]), o[0], 0, 0, 0);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[0][0]), sp_bf(ph[h2]), o[0], 0, 0, 0);
                    continue;
                }
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][2]), sp_bf(ph[h2]), o[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][0]), sp_bf(pl[h2]), o[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][1]), sp_bf(pm[h2]), o[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][1]), sp_bf(ph[h2]), o[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][0]), sp_bf(pm[h2]), o[dt], 0, 0, 0);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(vf[dt][0]), sp_bf(ph[h2]), o[dt], 0, 0, 0);
            }
        }
        if constexpr (!LG) l_run = l_run * alpha + psum;

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    if constexpr (PACK) {   // column 32 + (r & 3) + 4 hi = the V_l, V_m and V_h row groups of the packed tile, smallest first (all in this lane)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[NDT - 1][r] = (o[NDT - 1][r + 8] + o[NDT - 1][r + 4]) + o[NDT - 1][r];
    }

    if constexpr (LG) l_run = hi ? 0.f : o[NDT - 1][SUM_R];    // the ones column's row of O^T: the whole row sum, in the lane with hi = 0
    const float m_log2 = LG ? (t_begin < t_end ? -(NEGM1 ? negm1 : negm[0]) : -INFINITY) : m_run * cs;
    if (n_split > 1) {   // a key slice: unnormalised rows + (maximum in log2 units, row sum) for launch_attention_combine
        const float l_tot = sp_partner_sum(l_run);
        if (q_ok) {
            float* po = p.part_o + (((long long)blockIdx.z * p.n + b) * p.nq + qrow) * ((long long)p.n_head * D) + hh * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int dcol = 32 * dt + 8 * rq + 4 * hi;
                    if (dcol < D) *reinterpret_cast<f32x4*>(po + dcol) = f32x4{o[dt][4 * rq], o[dt][4 * rq + 1], o[dt][4 * rq + 2], o[dt][4 * rq + 3]};
                }
            if (hi == 0) *reinterpret_cast<f32x2*>(p.part_ml + ((((long long)blockIdx.z * p.n + b) * p.n_head + hh) * p.nq + qrow) * 2) = f32x2{m_log2, l_tot};
        }
        return;
    }
    const float inv = 1.0f / sp_partner_sum(l_run);
    if (q_ok) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int dcol = 32 * dt + 8 * rq + 4 * hi;
                if (dcol < D) {
                    const f32x4 w = {o[dt][4 * rq] * inv, o[dt][4 * rq + 1] * inv, o[dt][4 * rq + 2] * inv, o[dt][4 * rq + 3] * inv};
                    if (p.o3) s3_store4(reinterpret_cast<unsigned char*>(p.o3) + ((long long)b * p.nq + qrow) * p.ldo3, hh * D + dcol, w);
                    else *reinterpret_cast<f32x4*>(Of + (long long)qrow * p.ldo + dcol) = w;
                }
            }
        }
    }
}

template <int D, int NW, bool PK, bool LG>
static hipError_t launch_attn_split_d(const AttnParams& p, hipStream_t stream) {
    auto k = attn_split_kernel<D, NW, PK, LG>;
    const size_t lds = AttnSpCfg<D, NW>::LDS_BYTES;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    dim3 grid((p.nq + 32 * NW - 1) / (32 * NW), p.n * p.n_head, p.kv_splits > 1 ? p.kv_splits : 1);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_attn_split_any(const AttnParams& p, hipStream_t stream) {
    // widest workgroup that still gives every CU a workgroup (256 CUs), key slices included: the 8-wave form runs two waves per SIMD (the softmax of one beside
    // the matrix instructions of the other) and splits a K / V tile with half the work per thread -- per unit of work it is 1.5 x the 4-wave form
    // (profiles/r05l: 64 x 64, one sample: 185 us on 256 4-wave workgroups, two samples 142 us on 256 8-wave workgroups)
    const long long bh = (long long)p.n * p.n_head * (p.kv_splits > 1 ? p.kv_splits : 1);
    const bool w8 = (long long)((p.nq + 255) / 256) * bh >= 256;
    if constexpr (D == 40) {
        if (p.pack_tail & 1) {
            if (p.pack_tail & 2) return w8 ? launch_attn_split_d<D, 8, true, true>(p, stream) : launch_attn_split_d<D, 4, true, true>(p, stream);
            return w8 ? launch_attn_split_d<D, 8, true, false>(p, stream) : launch_attn_split_d<D, 4, true, false>(p, stream);
        }
    }
    if (p.pack_tail & 2) return w8 ? launch_attn_split_d<D, 8, false, true>(p, stream) : launch_attn_split_d<D, 4, false, true>(p, stream);
    // round 4's softmax (the A/B form, attn_pack_tail bit 1 = 0): its 8-wave d = 80 instantiation needs 260 registers (it spilled 4); that form runs 4-wave workgroups at d = 80
    if constexpr (D == 80) return launch_attn_split_d<D, 4, false, false>(p, stream);
    else return w8 ? launch_attn_split_d<D, 8, false, false>(p, stream) : launch_attn_split_d<D, 4, false, false>(p, stream);
}

bool attn_split_supported(const AttnParams& p) {
    return !p.bf16 && !p.mask && (p.d_head == 40 || p.d_head == 80) && !((p.ldq | p.ldk | p.ldv | p.ldo) & 3);
}

// fp32 q/k/v/o, no additive mask, d_head 40 or 80, 16-byte aligned rows (attn_split_supported)
hipError_t launch_attention_split(const AttnParams& p, hipStream_t stream) {
    if (!attn_split_supported(p)) return hipErrorInvalidValue;
    if (p.kv_splits > 1 && (!p.part_o || !p.part_ml)) return hipErrorInvalidValue;
    if (p.d_head == 40) return launch_attn_split_any<40>(p, stream);
    return launch_attn_split_any<80>(p, stream);
}

}  // namespace sdmi
