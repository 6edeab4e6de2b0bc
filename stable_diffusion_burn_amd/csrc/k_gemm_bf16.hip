// k_gemm_bf16.hip -- bf16 implicit-GEMM conv / linear (precision = 1: BASELINE.json configs[2..3]).
//
// Same kernel structure as k_gemm2.hip (buffer loads with hardware range check, XOR-swizzled
// unpadded LDS, in-loop software pipeline, swapped operands, XCD-aware tile map, deterministic
// split-K) with bf16 storage: a k tile is 64 bf16 = the same 128 BYTES per row as 32 fp32, so the
// staging / swizzle / fragment byte arithmetic is identical; a 16-byte fragment now holds 8
// consecutive k, which is exactly one operand of v_mfma_f32_16x16x32_bf16 (lane group g supplies
// k = 8g..8g+7).  Accumulation and the epilogue (bias + time-embedding row + residual) are fp32;
// the output is rounded to bf16 (round-to-nearest-even) or kept fp32 (`out_mode` = 1: the 4-channel
// eps prediction, the VAE's score matrix, the final RGB).  Channel slices are 64 wide
// (k = (cs*T + tap)*64 + ci): every hot-path layer with Cin >= 64 has Cin % 64 == 0; the three
// Cin = 4 layers stay on the fp32 kernel (which can emit bf16, `out_mode` = 2).
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned kOobB = 0xFFFFFFF0u;

__device__ __forceinline__ u32x4 buf_load16u(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
typedef float g16_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g16_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {   // one v_cvt_pk_bf16_f32 (round to nearest even, as f32_to_bf16_bits)
    const g16_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, g16_bf16x2));
}

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_bf16_kernel(const ConvGemm p) {
    constexpr bool GENERIC = false;
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int PA = (BM + 31) / 32;
    constexpr int PB = (BN + 31) / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMR = PA * 32;      // LDS rows (>= BM; rows beyond BM/BN are written but never read)
    constexpr int BNR = PB * 32;
    float* As = smem;                 // [2][BMR][32]
    float* Bs = smem + 2 * BMR * 32;  // [2][BNR][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int lid = gw.lid;
    const int tm = gw.tm;
    const int tn = gw.tn;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bt), 0, (int)p.b_bytes, 0x00020000);

    const int lrow = tid >> 3;
    const int kq = tid & 7;
    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const unsigned pix_bytes = (unsigned)p.a_ld * 2u;

    int a_iy0[PA], a_ix0[PA];
    unsigned a_nboff[PA];
    bool a_ok[PA];
#pragma unroll
    for (int pa = 0; pa < PA; ++pa) {
        const int r = pa * 32 + lrow;
        const int m = m0 + r;
        const bool ok = (r < BM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_ok[pa] = ok;
        a_nboff[pa] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * pix_bytes;
        a_iy0[pa] = oy * p.stride - p.pad;
        a_ix0[pa] = ox * p.stride - p.pad;
    }
    unsigned b_off[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int r = pb * 32 + lrow;
        const int n = n0 + r;
        const bool ok = (r < BN) && (n < p.N);
        b_off[pb] = ok ? ((unsigned)n * (unsigned)p.b_ld * 2u + (unsigned)kq * 16u) : kOobB;
    }

    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;  // next k tile gload() will fetch

    u32x4 ra[PA], rb[PB];

    auto gload = [&]() {
        const bool tile_ok = kt_next < kt_end;
        const unsigned k0b = (unsigned)kt_next * 128u;
        if constexpr (!GENERIC) {
            const unsigned c0b = (unsigned)(cs * 64 + kq * 8) * 2u;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                const int iy = a_iy0[pa] + ky;
                const int ix = a_ix0[pa] + kx;
                const bool ok = tile_ok & a_ok[pa] & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const unsigned pix = (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups));
                const unsigned off = a_nboff[pa] + pix * pix_bytes + c0b;
                ra[pa] = buf_load16u(rsA, ok ? off : kOobB);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const bool ok = tile_ok & (b_off[pb] != kOobB);
                rb[pb] = buf_load16u(rsB, ok ? b_off[pb] + k0b : kOobB);
            }
            const bool wrap_x = (kx + 1 == p.KW);
            const bool wrap_y = wrap_x && (ky + 1 == p.KH);
            kx = wrap_x ? 0 : kx + 1;
            ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
            cs = wrap_y ? cs + 1 : cs;
        } else {
            const int k = kt_next * 32 + kq * 4;
            const bool kok = tile_ok && (k < p.K);
            const int kk = kok ? k : 0;
            const int sl = kk / p.CS;
            const int ci = kk - sl * p.CS;
            const int gcs = sl / T;
            const int gtap = sl - gcs * T;
            const int gky = gtap / p.KW;
            const int gkx = gtap - gky * p.KW;
            const unsigned c0b = (unsigned)(gcs * p.CS + ci) * 4u;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                const int iy = a_iy0[pa] + gky;
                const int ix = a_ix0[pa] + gkx;
                const bool ok = kok & a_ok[pa] & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const unsigned pix = (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups));
                const unsigned off = a_nboff[pa] + pix * pix_bytes + c0b;
                ra[pa] = buf_load16u(rsA, ok ? off : kOobB);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const bool ok = kok & (b_off[pb] != kOobB);
                rb[pb] = buf_load16u(rsB, ok ? b_off[pb] + k0b : kOobB);
            }
        }
        ++kt_next;
    };

    // LDS addressing: row r, 16-byte chunk q -> float offset r*32 + ((q ^ (r & 7)) << 2)
    const int st_chunk = (kq ^ (lrow & 7)) << 2;  // rows pa*32 + lrow: (row & 7) == (lrow & 7)
    auto lstore = [&](int buf) {
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int r = pa * 32 + lrow;
            *reinterpret_cast<u32x4*>(As + (buf * BMR + r) * 32 + st_chunk) = ra[pa];
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int r = pb * 32 + lrow;
            *reinterpret_cast<u32x4*>(Bs + (buf * BNR + r) * 32 + st_chunk) = rb[pb];
        }
    };

    // fragment reads: lane (c = lane&15, g = lane>>4) reads row base+c, chunk kk*4+g (swizzled by row&7 = c&7,
    // tile row bases are multiples of 16)
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr_chunk0 = ((0 + g4) ^ (c15 & 7)) << 2;
    const int fr_chunk1 = ((4 + g4) ^ (c15 & 7)) << 2;
    const float* a_row = As + (wm * 16 * MI + c15) * 32;
    const float* b_row = Bs + (wn * 16 * NI + c15) * 32;
    auto lread = [&](int buf, int kk, u32x4 (&a)[MI], u32x4 (&b)[NI]) {
        const int ch = kk ? fr_chunk1 : fr_chunk0;
        const float* ab = a_row + buf * BMR * 32 + ch;
        const float* bb = b_row + buf * BNR * 32 + ch;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const u32x4*>(ab + mi * 16 * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const u32x4*>(bb + ni * 16 * 32);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one v_mfma_f32_16x16x32_bf16 per (mi, ni) per 32-wide k half; rows mi0..mi1-1 of the wave tile
    auto mma = [&](const u32x4 (&a)[MI], const u32x4 (&b)[NI], int mi0, int mi1) {
#pragma unroll
        for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[ni]), __builtin_bit_cast(bf16x8, a[mi]),
                                                                      acc[mi][ni], 0, 0, 0);
    };

    u32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];

    gload();        // tile 0
    lstore(0);
    gload();        // tile 1 (or zeros)
    __syncthreads();
    lread(0, 0, fa0, fb0);

    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        lread(cur, 1, fa1, fb1);
        mma(fa0, fb0, 0, MI);
        lstore(cur ^ 1);   // tile t+1 (zeros past the end)
        gload();           // tile t+2 into the just-freed staging registers
        mma(fa1, fb1, 0, (MI + 1) / 2);
        __syncthreads();
        lread(cur ^ 1, 0, fa0, fb0);
        mma(fa1, fb1, (MI + 1) / 2, MI);
    }

    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual, then bf16 or fp32 store ----------
    const bool split = p.splits > 1;
    const bool vec_ok = ((p.N & 3) == 0) && (((split ? p.N : p.ldc) & 3) == 0) && ((p.ldr & 3) == 0 || !p.resid);
    const bool out_f32 = split || p.out_mode == 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const int ldc = split ? p.N : p.ldc;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        if (m >= p.M) continue;
        const int smp = m / HoWo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 16 + g4 * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[mi][ni];
            if (vec_ok) {
                if (!split) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    if (p.resid) {
                        const u32x2 r = *reinterpret_cast<const u32x2*>(Rh + (long long)m * p.ldr + n);
                        v[0] += bf16_lo(r[0]); v[1] += bf16_hi(r[0]); v[2] += bf16_lo(r[1]); v[3] += bf16_hi(r[1]);
                    }
                }
                if (split) {
                    *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;     // (Cf = this k slice's slab)
                } else if (out_f32) {
                    *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
                } else {
                    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(Ch + (long long)m * ldc + n) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < p.N) {
                        float s = v[r];
                        if (!split) {
                            if (p.bias) s += p.bias[n + r];
                            if (p.rowvec) s += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                            if (p.resid) s += __uint_as_float((unsigned)Rh[(long long)m * p.ldr + n + r] << 16);
                        }
                        if (out_f32) Cf[(long long)m * ldc + n + r] = s;
                        else Ch[(long long)m * ldc + n + r] = (unsigned short)f32_to_bf16_bits(s);
                    }
                }
            }
        }
    }
}

// ---- split-K reduction for the bf16 path: fp32 slabs -> bf16 (or fp32) output ---------------------------
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const ConvGemm p) {
    const float* slabs = p.slabs;
    const int HoWo = p.Ho * p.Wo;
    const bool out_f32 = p.out_mode == 1;
    float* Cf = p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const long long total = (long long)p.M * p.N;
    // round 6: four outputs per thread (16-byte slab loads, two slices in flight), the bias / time-embedding row / residual requested in front of the slab loads;
    // the same sums in the same order as the scalar form below (which stays for odd strides): bit-identical
    if (((p.N | p.ldc | p.rowvec_stride) & 3) == 0 && (!p.resid || (p.ldr & 3) == 0)) {
        typedef float rf32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int ru32x2 __attribute__((ext_vector_type(2)));
        const int n4 = p.N >> 2;
        const long long total4 = (long long)p.M * n4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
            const int m = (int)(i / n4);
            const int n = (int)(i - (long long)m * n4) * 4;
            const long long off = (long long)m * p.N + n;
            rf32x4 eb = {0.f, 0.f, 0.f, 0.f}, ev = eb, er = eb;
            if (p.bias) eb = *reinterpret_cast<const rf32x4*>(p.bias + n);
            if (p.rowvec) ev = *reinterpret_cast<const rf32x4*>(p.rowvec + (long long)(m / HoWo) * p.rowvec_stride + n);
            if (p.resid) {
                const ru32x2 rr = *reinterpret_cast<const ru32x2*>(Rh + (long long)m * p.ldr + n);
                er = rf32x4{__uint_as_float(rr[0] << 16), __uint_as_float(rr[0] & 0xFFFF0000u), __uint_as_float(rr[1] << 16), __uint_as_float(rr[1] & 0xFFFF0000u)};
            }
            rf32x4 v = *reinterpret_cast<const rf32x4*>(slabs + off);
            int sl = 1;
            for (; sl + 1 < p.splits; sl += 2) {
                const rf32x4 a0 = *reinterpret_cast<const rf32x4*>(slabs + (long long)sl * p.slab_stride + off);
                const rf32x4 a1 = *reinterpret_cast<const rf32x4*>(slabs + (long long)(sl + 1) * p.slab_stride + off);
                v += a0; v += a1;
            }
            if (sl < p.splits) v += *reinterpret_cast<const rf32x4*>(slabs + (long long)sl * p.slab_stride + off);
            if (p.bias) v += eb;
            if (p.rowvec) v += ev;
            if (p.resid) v += er;
            if (out_f32) *reinterpret_cast<rf32x4*>(Cf + (long long)m * p.ldc + n) = v;
            else {
                const ru32x2 o = {f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16), f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16)};
                *reinterpret_cast<ru32x2*>(Ch + (long long)m * p.ldc + n) = o;
            }
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / p.N);
        const int n = (int)(i - (long long)m * p.N);
        float v = slabs[i];
        for (int s = 1; s < p.splits; ++s) v += slabs[s * p.slab_stride + i];
        if (p.bias) v += p.bias[n];
        if (p.rowvec) v += p.rowvec[(long long)(m / HoWo) * p.rowvec_stride + n];
        if (p.resid) v += __uint_as_float((unsigned)Rh[(long long)m * p.ldr + n] << 16);
        if (out_f32) Cf[(long long)m * p.ldc + n] = v;
        else Ch[(long long)m * p.ldc + n] = (unsigned short)f32_to_bf16_bits(v);
    }
}

// ---- weight packing to bf16 (channel slices of 64) ---------------------------------------------------------
__global__ void pack_conv_weight_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ bt, int cout, int cin, int kh, int kw) {
    const int T = kh * kw;
    const int CS = 64;
    const long long K = (long long)cin * T;
    const long long total = (long long)cout * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K);
        const int k = (int)(i - (long long)n * K);
        const int sl = k / CS;
        const int ci = k - sl * CS;
        const int cs = sl / T;
        const int tap = sl - cs * T;
        const int c = cs * CS + ci;
        bt[i] = (unsigned short)f32_to_bf16_bits(w[((long long)n * cin + c) * T + tap]);
    }
}

__global__ void pack_linear_weight_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ bt, int cin, int cout) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ci = by + r, co = bx + tx;
        tile[r][tx] = (ci < cin && co < cout) ? w[(long long)ci * cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = bx + r, ci = by + tx;
        if (co < cout && ci < cin) bt[(long long)co * cin + ci] = (unsigned short)f32_to_bf16_bits(tile[tx][r]);
    }
}

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg_bf16(const ConvGemm& p, size_t lds, dim3 grid, hipStream_t stream) {
    auto k = conv_gemm_bf16_kernel<MI, NI, WM, WN>;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv_gemm_bf16(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTiles) return hipErrorInvalidValue;
    if (p.Cin % 64) return hipErrorInvalidValue;
    const int bm = gemm_tile_info(cfg).bm, bn = gemm_tile_info(cfg).bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bn - 1) / bn;
    const int tiles = MT * NT;
    const dim3 grid = gemm_grid(p, tiles);
    const size_t lds = gemm2_tile_lds_bytes(cfg);
    switch (cfg) {
        case 0: return launch_cfg_bf16<4, 4, 2, 2>(p, lds, grid, stream);
        case 1: return launch_cfg_bf16<4, 2, 2, 2>(p, lds, grid, stream);
        case 2: return launch_cfg_bf16<2, 2, 2, 2>(p, lds, grid, stream);
        case 3: return launch_cfg_bf16<8, 4, 2, 2>(p, lds, grid, stream);
        case 4: return launch_cfg_bf16<2, 5, 4, 1>(p, lds, grid, stream);
        case 5: return launch_cfg_bf16<4, 5, 4, 1>(p, lds, grid, stream);
        case 6: return launch_cfg_bf16<2, 4, 2, 2>(p, lds, grid, stream);
        case 7: return launch_cfg_bf16<4, 5, 2, 2>(p, lds, grid, stream);
        case 8: return launch_cfg_bf16<1, 5, 4, 1>(p, lds, grid, stream);
        case 9: return launch_cfg_bf16<2, 5, 2, 2>(p, lds, grid, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_splitk_reduce_bf16(const ConvGemm& p, hipStream_t stream) {
    const bool vec = ((p.N | p.ldc | p.rowvec_stride) & 3) == 0 && (!p.resid || (p.ldr & 3) == 0);    // the kernel's 16-byte path: four outputs per thread
    const long long work = (long long)p.M * p.N / (vec ? 4 : 1);
    int blocks = (int)((work + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_pack_conv_weight_bf16(const float* w, void* bt, int cout, int cin, int kh, int kw, hipStream_t s) {
    const long long total = (long long)cout * cin * kh * kw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv_weight_bf16_kernel, dim3(blocks), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(bt), cout, cin, kh, kw);
    return hipGetLastError();
}

hipError_t launch_pack_linear_weight_bf16(const float* w, void* bt, int cin, int cout, hipStream_t s) {
    dim3 grid((cout + 31) / 32, (cin + 31) / 32);
    hipLaunchKernelGGL(pack_linear_weight_bf16_kernel, grid, dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(bt), cin, cout);
    return hipGetLastError();
}

}  // namespace sdmi
