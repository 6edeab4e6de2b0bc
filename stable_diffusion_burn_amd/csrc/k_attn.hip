// k_attn.hip -- flash-style attention on the gfx950 fp32 matrix cores.
//
// Replaces qkv_attention (reference src/model/attention.rs:5-45, byte-identical
// copy in src/backend.rs:88-128): softmax((q*s)(k*s)^T + mask) v per head, with
// s = d_head^-0.25 applied to q and to k as the reference does.  The reference
// materialises the [n, heads, Nq, Nk] score tensor (537 MB at 64x64); here the
// scores never leave registers.
//
// MI355X mapping (one 256-thread workgroup = 4 waves = 64 query rows of one
// (batch, head); each wave owns 16 query rows):
//  * S^T tile = K Q^T on v_mfma_f32_16x16x4_f32 with the key tile as the MFMA A
//    operand and the (register-resident, pre-scaled) Q fragment as B, so lane
//    (g = lane>>4, c = lane&15) holds S[key = 4g+r][query = c]: a query's scores
//    live in 4 lanes x 4 registers, the row max needs two xor-shuffles and the
//    row sum stays lane-local until the end.
//  * P never goes through LDS: in that layout the exponentiated scores ARE the
//    MFMA B operand of O^T = V^T P^T (hardware k index g at step r <-> key 4g+r),
//    and the V^T fragment is a ds_read_b32 column of the [key][d] LDS tile.
//    O^T accumulators hold O[query = c][d = 4g..4g+3] -> the online-softmax
//    rescale is a per-lane scalar multiply and the output store is 16 bytes.
//  * K/V tiles are staged global -> registers -> LDS (row stride d+4 floats:
//    conflict-free b64 K reads and b32 V reads), double buffered, one barrier
//    per KV tile; the next tile's loads are issued before the current MFMAs.
//  * head dims 40 / 80 / 160 (SD v1.4 UNet) are template instances; d = 40 pads
//    the PV output to 48 columns (third 16-wide tile half used), QK^T uses K = d
//    exactly (k steps of 4).
#include "kernels.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2h __attribute__((ext_vector_type(2)));

// =====================================================================================
// 4- or 8-wave workgroups, branch-free fast path, exp2 domain, permlane reductions
// =====================================================================================
// What the ISA + rocprof of the first (4-wave, shuffle-based) version of this kernel led to:
//  * NW = 8 waves (128 query rows) per workgroup for long sequences: the 64x64 level has
//    1024 4-wave workgroups for 768 resident slots (LDS-limited) -> 1.33 "rounds", a 33 %
//    quantisation loss; 512 8-wave workgroups are all resident at once (4 waves/SIMD) and
//    the K/V staging traffic per MFMA halves.
//  * the additive mask is a template flag and key-length masking runs only on the last
//    (partial) tile: v1 carried 16 exec-mask branches per tile even with mask == NULL.
//  * scores are kept in log2 units (q pre-scaled by d^-0.5 * log2 e, K unscaled):
//    exp(s - m) becomes one v_exp_f32, no multiply.
//  * the two cross-lane reductions use v_permlane32_swap / v_permlane16_swap (VALU) instead
//    of ds_bpermute round trips through the LDS crossbar.
// The swaps are issued as inline asm: with ROCm 7.2's hipcc the __builtin_amdgcn_permlane{16,32}_swap
// results fed to fmaxf / fadd get folded by InstCombine into a single extractvalue (the exchange is
// silently dropped; found by the parity tests).  "s_nop 1" covers the VALU-write -> permlane-read
// hazard (2 wait states) that hipcc does not pad inside an asm statement.
__device__ __forceinline__ void swap32(float a, float b, float& ra, float& rb) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    ra = a;
    rb = b;
}
__device__ __forceinline__ void swap16(float a, float b, float& ra, float& rb) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    ra = a;
    rb = b;
}
__device__ __forceinline__ float xlane_max4(float v) {  // max over lanes {c, c+16, c+32, c+48}
    float a, b;
    swap32(v, v, a, b);   // a = [lo, lo], b = [hi, hi]
    v = fmaxf(a, b);
    swap16(v, v, a, b);   // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
    return fmaxf(a, b);
}
__device__ __forceinline__ float xlane_sum4(float v) {
    float a, b;
    swap32(v, v, a, b);
    v = a + b;
    swap16(v, v, a, b);
    return a + b;
}

template <int D, int NW, bool H16 = false>
struct Attn2Cfg {
    static constexpr int NT = NW * 64;
    static constexpr int BKV = (D > 96) ? 32 : 64;
    static constexpr int KT = BKV / 16;
    static constexpr int DT = (D + 15) / 16;
    static constexpr int DC = D / 8;
    static constexpr int LDK = D + 4;
    static constexpr int VEC = H16 ? 8 : 4;                 // elements per 16-byte global access
    static constexpr int F4_PER_TILE = BKV * (D / VEC);       // 16-byte vectors per K (or V) tile
    static constexpr int NLD = (F4_PER_TILE + NT - 1) / NT;
    static constexpr int TILE_FLOATS = BKV * LDK;
    static constexpr size_t LDS_BYTES = (size_t)(4 * TILE_FLOATS + 64) * sizeof(float);
};

// H16: q/k/v/o are bf16 in HBM (precision = 1); the tiles are widened to fp32 when staged into LDS and
// all arithmetic (QK^T, softmax, PV) stays on the fp32 matrix path.
template <int D, int NW, bool HAS_MASK, bool H16>
__global__ __launch_bounds__(NW * 64) void attn2_kernel(const AttnParams p) {
    using Cfg = Attn2Cfg<D, NW, H16>;
    constexpr int VEC = Cfg::VEC;
    typedef unsigned short h16;
    constexpr int NT = Cfg::NT, BKV = Cfg::BKV, KT = Cfg::KT, DT = Cfg::DT, DC = Cfg::DC, LDK = Cfg::LDK, NLD = Cfg::NLD;
    constexpr float kLog2e = 1.4426950408889634f;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + 2 * Cfg::TILE_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;
    const int c = lane & 15;

    const int b = blockIdx.y / p.n_head;
    const int hh = blockIdx.y - b * p.n_head;
    const int qrow = blockIdx.x * (16 * NW) + wave * 16 + c;
    const bool q_ok = qrow < p.nq;

    const float* Qb = p.q + (long long)b * p.q_bs + hh * D;
    const float* Kb = p.k + (long long)b * p.k_bs + hh * D;
    const float* Vb = p.v + (long long)b * p.v_bs + hh * D;
    float* Ob = p.o + (long long)b * p.o_bs + hh * D;
    const h16* Qh = reinterpret_cast<const h16*>(p.q) + (long long)b * p.q_bs + hh * D;
    const h16* Kh = reinterpret_cast<const h16*>(p.k) + (long long)b * p.k_bs + hh * D;
    const h16* Vh = reinterpret_cast<const h16*>(p.v) + (long long)b * p.v_bs + hh * D;
    h16* Oh = reinterpret_cast<h16*>(p.o) + (long long)b * p.o_bs + hh * D;

    const int nk = p.kv_len ? p.kv_len[b] : p.nk;
    const int n_tiles = (nk + BKV - 1) / BKV;
    const int n_full = nk / BKV;
    // key slices (kv_splits > 1; fp32, no mask): this workgroup's tiles [t_begin, t_end)
    const int n_split = (!H16 && !HAS_MASK && p.kv_splits > 1) ? p.kv_splits : 1;
    const int tps = (n_tiles + n_split - 1) / n_split;
    const int t_begin = min((int)blockIdx.z * tps, n_tiles), t_end = min(t_begin + tps, n_tiles);


    f32x4 rk[NLD], rv[NLD];
    auto gload = [&](int tile) {
        const int kv0 = tile * BKV;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / (D / VEC);
            const int c4 = idx - row * (D / VEC);
            const int key = kv0 + row;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};  // H16: 8 raw bf16 each
            if (idx < Cfg::F4_PER_TILE && key < nk) {
                if constexpr (H16) {
                    kk = *reinterpret_cast<const f32x4*>(Kh + (long long)key * p.ldk + c4 * 8);
                    vv = *reinterpret_cast<const f32x4*>(Vh + (long long)key * p.ldv + c4 * 8);
                } else {
                    kk = *reinterpret_cast<const f32x4*>(Kb + (long long)key * p.ldk + c4 * 4);
                    vv = *reinterpret_cast<const f32x4*>(Vb + (long long)key * p.ldv + c4 * 4);
                }
            }
            rk[i] = kk;
            rv[i] = vv;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * NT;
            if (idx < Cfg::F4_PER_TILE) {
                const int row = idx / (D / VEC);
                const int c4 = idx - row * (D / VEC);
                float* kd = Ks + buf * Cfg::TILE_FLOATS + row * LDK + c4 * VEC;
                float* vd = Vs + buf * Cfg::TILE_FLOATS + row * LDK + c4 * VEC;
                const bool sw = (row & 8) != 0;   // see the K fragment read
                if constexpr (H16) {
                    const u32x4h kw = __builtin_bit_cast(u32x4h, rk[i]), vw = __builtin_bit_cast(u32x4h, rv[i]);
                    const f32x4 k0 = {__uint_as_float(kw[0] << 16), __uint_as_float(kw[0] & 0xFFFF0000u),
                                      __uint_as_float(kw[1] << 16), __uint_as_float(kw[1] & 0xFFFF0000u)};
                    const f32x4 k1 = {__uint_as_float(kw[2] << 16), __uint_as_float(kw[2] & 0xFFFF0000u),
                                      __uint_as_float(kw[3] << 16), __uint_as_float(kw[3] & 0xFFFF0000u)};
                    *reinterpret_cast<f32x4*>(kd) = sw ? f32x4{k0[2], k0[3], k0[0], k0[1]} : k0;
                    *reinterpret_cast<f32x4*>(kd + 4) = sw ? f32x4{k1[2], k1[3], k1[0], k1[1]} : k1;
                    *reinterpret_cast<f32x4*>(vd) = f32x4{__uint_as_float(vw[0] << 16), __uint_as_float(vw[0] & 0xFFFF0000u),
                                                          __uint_as_float(vw[1] << 16), __uint_as_float(vw[1] & 0xFFFF0000u)};
                    *reinterpret_cast<f32x4*>(vd + 4) = f32x4{__uint_as_float(vw[2] << 16), __uint_as_float(vw[2] & 0xFFFF0000u),
                                                              __uint_as_float(vw[3] << 16), __uint_as_float(vw[3] & 0xFFFF0000u)};
                } else {
                    *reinterpret_cast<f32x4*>(kd) = sw ? f32x4{rk[i][2], rk[i][3], rk[i][0], rk[i][1]} : rk[i];
                    *reinterpret_cast<f32x4*>(vd) = rv[i];
                }
            }
        }
    };

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY;
    float l_run = 0.f;

    // (round 6: the first K / V tile is requested before the query row, k_attn_split.hip)
    if (t_begin < t_end) gload(t_begin);
    const float qscale = p.scale * p.scale * kLog2e;  // (q*s)(k*s) = q k s^2, in log2 units
    f32x2 qf[DC];
#pragma unroll
    for (int cc = 0; cc < DC; ++cc) {
        f32x2 v = {0.f, 0.f};
        if (q_ok) {
            if constexpr (H16) {
                const unsigned w = *reinterpret_cast<const unsigned*>(Qh + (long long)qrow * p.ldq + cc * 8 + g * 2);
                v = f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)};
            } else {
                v = *reinterpret_cast<const f32x2*>(Qb + (long long)qrow * p.ldq + cc * 8 + g * 2);
            }
        }
        qf[cc] = v * qscale;
    }
    if (t_begin < t_end) lstore(0);
    __syncthreads();

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int cur = (tile - t_begin) & 1;
        const bool more = (tile + 1) < t_end;
        if (more) gload(tile + 1);

        const float* Kt = Ks + cur * Cfg::TILE_FLOATS;
        const float* Vt = Vs + cur * Cfg::TILE_FLOATS;
        const int kv0 = tile * BKV;

        f32x4 s[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            // K rows 8..15 of every 16 are stored with the two halves of each 4-float chunk swapped (lstore): hipcc merges
            // these loads into ds_read2_b64, which banks 16 lanes at a time modulo 32 dwords, where rows c and c + 8
            // (8 LDK = 0 mod 32 for any 16-byte-aligned row stride) would otherwise collide 2-way (measured: 8 % of CU cycles)
            const float* kp = Kt + (kt * 16 + c) * LDK + ((g * 2) ^ ((c >> 3) << 1));
#pragma unroll
            for (int cc = 0; cc < DC; ++cc) {
                const f32x2 kf = *reinterpret_cast<const f32x2*>(kp + cc * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0], qf[cc][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[1], qf[cc][1], a, 0, 0, 0);
            }
            s[kt] = a;
        }

        if (HAS_MASK || tile >= n_full) {  // uniform: additive mask and/or the ragged last tile
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + kt * 16 + g * 4 + r;
                    float v = s[kt][r];
                    if (HAS_MASK) {
                        if (q_ok && key < nk) v += p.mask[(long long)qrow * p.mask_ld + key] * kLog2e;
                    }
                    if (key >= nk) v = -INFINITY;
                    s[kt][r] = v;
                }
            }
        }

        float mt = s[0][0];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[kt][r]);
        mt = xlane_max4(mt);
        const float m_new = fmaxf(m_run, mt);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - m_use);
                s[kt][r] = e;
                psum += e;
            }
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= alpha;

#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = Vt + (kt * 16 + g * 4 + r) * LDK + c;
                const float pv = s[kt][r];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[dt * 16], pv, o[dt], 0, 0, 0);
            }
        }

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    if constexpr (!H16 && !HAS_MASK) {
        if (n_split > 1) {   // a key slice: unnormalised rows + (maximum in log2 units, row sum) for launch_attention_combine
            const float l_tot = xlane_sum4(l_run);
            if (q_ok) {
                float* po = p.part_o + (((long long)blockIdx.z * p.n + b) * p.nq + qrow) * ((long long)p.n_head * D) + hh * D;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int dcol = dt * 16 + g * 4;
                    if (dcol < D) *reinterpret_cast<f32x4*>(po + dcol) = o[dt];
                }
                if (g == 0) *reinterpret_cast<f32x2*>(p.part_ml + ((((long long)blockIdx.z * p.n + b) * p.n_head + hh) * p.nq + qrow) * 2) = f32x2{m_run, l_tot};
            }
            return;
        }
    }
    const float inv = 1.0f / xlane_sum4(l_run);
    if (q_ok) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int dcol = dt * 16 + g * 4;
            if (dcol < D) {
                const f32x4 r = o[dt] * inv;
                if constexpr (H16) {
                    auto bits = [](float f) { unsigned u = __float_as_uint(f); u += 0x7FFFu + ((u >> 16) & 1u); return u >> 16; };
                    u32x2h w = {bits(r[0]) | (bits(r[1]) << 16), bits(r[2]) | (bits(r[3]) << 16)};
                    *reinterpret_cast<u32x2h*>(Oh + (long long)qrow * p.ldo + dcol) = w;
                } else {
                    if (p.o3) s3_store4(reinterpret_cast<unsigned char*>(p.o3) + ((long long)b * p.nq + qrow) * p.ldo3, hh * D + dcol, r);
                    else *reinterpret_cast<f32x4*>(Ob + (long long)qrow * p.ldo + dcol) = r;
                }
            }
        }
    }
}

// ---- row softmax for the unfused single-head (VAE, d = 512) attention ------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* x, int rows, int cols, float scale) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    if (row >= rows) return;
    float* xr = x + (long long)row * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, xr[i] * scale);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < cols; i += 256) {
        const float e = __expf(xr[i] * scale - mx);
        xr[i] = e;
        sum += e;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    for (int i = tid; i < cols; i += 256) xr[i] *= inv;
}

// ---- host side ------------------------------------------------------------------------------
bool attn_supported_head_dim(int d) { return d == 40 || d == 64 || d == 80 || d == 160; }  // 64: CLIP (768 / 12)

template <int D, int NW, bool HAS_MASK, bool H16>
static hipError_t launch_attn2_d(const AttnParams& p, hipStream_t stream) {
    auto k = attn2_kernel<D, NW, HAS_MASK, H16>;
    const size_t lds = Attn2Cfg<D, NW, H16>::LDS_BYTES;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    dim3 grid((p.nq + 16 * NW - 1) / (16 * NW), p.n * p.n_head, (!H16 && !HAS_MASK && p.kv_splits > 1) ? p.kv_splits : 1);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_attn2_any(const AttnParams& p, hipStream_t stream) {
    // 8-wave workgroups once they still fill the chip (>= 1.5 workgroups per CU), else 4-wave
    const long long wg8 = (long long)((p.nq + 127) / 128) * p.n * p.n_head;
    const bool big = wg8 >= 384;
    if (p.bf16) {
        if (p.mask) return hipErrorInvalidValue;  // the masked (CLIP) path is fp32 only
        return big ? launch_attn2_d<D, 8, false, true>(p, stream) : launch_attn2_d<D, 4, false, true>(p, stream);
    }
    if (p.mask) return big ? launch_attn2_d<D, 8, true, false>(p, stream) : launch_attn2_d<D, 4, true, false>(p, stream);
    return big ? launch_attn2_d<D, 8, false, false>(p, stream) : launch_attn2_d<D, 4, false, false>(p, stream);
}

// ---- merge of the key slices of a kv_splits > 1 launch (this file's fp32 kernel or k_attn_split.hip's) -----------------------------------------
// One thread per (sample, query, head, 4 output channels): m = max_s m_s, w_s = 2^(m_s - m), out = sum_s w_s o_s / sum_s w_s l_s, slices in order (deterministic);
// a slice without keys has m_s = -inf, l_s = 0 and drops out.  Writes fp32 rows or the three bf16 planes, like the kernels' own epilogues.
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnParams p) {
    const int D = p.d_head, d4 = D >> 2;
    const long long total = (long long)p.n * p.nq * p.n_head * d4;
    const long long row_elems = (long long)p.n_head * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % d4);
        const int hh = (int)((i / d4) % p.n_head);
        const long long bq = i / ((long long)d4 * p.n_head);
        const int q = (int)(bq % p.nq), b = (int)(bq / p.nq);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
        if (p.kv_splits <= 8) {
            // round 6: every slice's (m, l) and output piece requested at once (<= 8 slices: 40 registers) -- the loops below are two to three DEPENDENT round trips per slice
            // (660 launches of ~ 8 us per batch-1 image were that chain).  Same operations in slice order: bit-identical.
            f32x2 ml[8];
            f32x4 po[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                ml[s] = f32x2{-INFINITY, 0.f};
                po[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < p.kv_splits) {
                    ml[s] = *reinterpret_cast<const f32x2*>(p.part_ml + ((((long long)s * p.n + b) * p.n_head + hh) * p.nq + q) * 2);
                    po[s] = *reinterpret_cast<const f32x4*>(p.part_o + (((long long)s * p.n + b) * p.nq + q) * row_elems + hh * D + c4 * 4);
                }
            }
            float m = -INFINITY;
#pragma unroll
            for (int s = 0; s < 8; ++s) m = fmaxf(m, ml[s][0]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < p.kv_splits && ml[s][0] != -INFINITY) {
                    const float w = __builtin_amdgcn_exp2f(ml[s][0] - m);
                    acc += po[s] * w;
                    l += ml[s][1] * w;
                }
            }
        } else {
            float m = -INFINITY;
            for (int s = 0; s < p.kv_splits; ++s) m = fmaxf(m, p.part_ml[((((long long)s * p.n + b) * p.n_head + hh) * p.nq + q) * 2]);
            for (int s = 0; s < p.kv_splits; ++s) {
                const f32x2 ml = *reinterpret_cast<const f32x2*>(p.part_ml + ((((long long)s * p.n + b) * p.n_head + hh) * p.nq + q) * 2);
                if (ml[0] == -INFINITY) continue;
                const float w = __builtin_amdgcn_exp2f(ml[0] - m);
                acc += *reinterpret_cast<const f32x4*>(p.part_o + (((long long)s * p.n + b) * p.nq + q) * row_elems + hh * D + c4 * 4) * w;
                l += ml[1] * w;
            }
        }
        const f32x4 r = acc * (1.0f / l);
        if (p.o3) s3_store4(reinterpret_cast<unsigned char*>(p.o3) + ((long long)b * p.nq + q) * p.ldo3, hh * D + c4 * 4, r);
        else *reinterpret_cast<f32x4*>(p.o + (long long)b * p.o_bs + (long long)q * p.ldo + hh * D + c4 * 4) = r;
    }
}

hipError_t launch_attention_combine(const AttnParams& p, hipStream_t stream) {
    if (p.kv_splits < 2 || !p.part_o || !p.part_ml || p.bf16 || (p.d_head & 3) || (!p.o3 && (p.ldo & 3))) return hipErrorInvalidValue;
    const long long total = (long long)p.n * p.nq * p.n_head * (p.d_head >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

int attn_f32_kv_tile(const AttnParams& p) { return attn_split_supported(p) ? 64 : (p.d_head > 96 ? 32 : 64); }

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (p.kv_splits > 1 && (p.bf16 || p.mask || !p.part_o || !p.part_ml)) return hipErrorInvalidValue;
    switch (p.d_head) {
        case 40: return launch_attn2_any<40>(p, stream);
        case 64: return launch_attn2_any<64>(p, stream);
        case 80: return launch_attn2_any<80>(p, stream);
        case 160: return launch_attn2_any<160>(p, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_softmax_rows(float* x, int rows, int cols, float scale, hipStream_t stream) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, rows, cols, scale);
    return hipGetLastError();
}

}  // namespace sdmi
