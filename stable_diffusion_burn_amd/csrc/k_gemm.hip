// k_gemm.hip -- implicit-GEMM convolution / linear on the gfx950 fp32 matrix cores.
//
// Replaces Burn's Conv2d::forward and Linear::forward at every call site of the
// hot path (reference src/model/unet/mod.rs:116-118,140,397,425,468,479,553,580,
// 645-651,716,719,726,729; src/model/autoencoder/mod.rs:69,206,214,319,516-523,
// 568-604).  One kernel template covers 3x3 s1/s2 p1 (+ virtual nearest-2x
// upsample of the source), 1x1 and plain GEMM (KH=KW=1):
//
//   C[m][n] = sum_k A(m,k) Bt[n][k] + bias[n] + rowvec[sample(m)][n] + resid[m][n]
//
// MI355X mapping
//  * v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/issue/SIMD, 157 TF chip peak).
//    Operands are used "swapped": the MFMA A operand is the weight tile (rows n),
//    the B operand is the activation tile (cols m), so each lane ends up holding
//    C[m = lane&15][n = 4*(lane>>4) .. +3]: four consecutive output channels ->
//    16-byte epilogue loads/stores of bias / residual / output.
//  * K permutation: within a 16-wide k chunk lane group g = lane>>4 supplies
//    k = 4g+j at MFMA step j for BOTH operands, so every fragment is a single
//    ds_read_b128 of 4 consecutive k (dot products do not care about k order).
//  * 256-thread workgroup = 4 waves (one per SIMD), wave tile (16*MI) x (16*NI).
//    LDS tiles [rows][32+4] fp32 (row stride 144 B: conflict-free b128 reads),
//    double buffered; global->register prefetch of k-tile t+1 is issued before
//    the MFMAs of tile t and written to LDS after them (one barrier per k tile).
//  * k order = (channel slice, tap, 32 channels): the 9 taps of a slice re-read
//    the same 128-byte pixel segments, which stay in L1/L2.
//  * blockIdx -> tile map is XCD aware: blocks that land on one XCD (bid % 8)
//    own a contiguous range of tiles (n fastest), so an XCD's L2 keeps its band
//    of the activation and its neighbours' halo rows.
//  * split-K over blockIdx.z writes raw fp32 slabs; launch_splitk_reduce sums
//    them in fixed order (bit-reproducible) and applies the epilogue.
#include "kernels.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLdsLd = 36;  // floats per LDS tile row (32 + 4 pad)

template <int MI, int NI, int WM, int WN, bool GENERIC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemm p) {
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int PA = (BM + 31) / 32;
    constexpr int PB = (BN + 31) / 32;
    constexpr int LD = kLdsLd;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                // [2][BM][LD]
    float* Bs = smem + 2 * BM * LD;  // [2][BN][LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    // ---- XCD-aware tile mapping (gridDim.x is a multiple of 8) -----------------
    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    const int tpx = gridDim.x >> 3;
    const int lid = (blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
    if (lid >= MT * NT) return;
    const int tm = lid / NT;
    const int tn = lid - tm * NT;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int z = blockIdx.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);

    // ---- per-thread gather state ------------------------------------------------
    const int lrow = tid >> 3;  // 0..31
    const int kq = tid & 7;     // float4 column within the 32-wide k tile
    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;

    int a_iy0[PA], a_ix0[PA], a_nb[PA];
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
        a_nb[pa] = nb;
        a_iy0[pa] = oy * p.stride - p.pad;
        a_ix0[pa] = ox * p.stride - p.pad;
    }
    bool b_ok[PB];
    const float* b_ptr[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int r = pb * 32 + lrow;
        const int n = n0 + r;
        b_ok[pb] = (r < BN) && (n < p.N);
        b_ptr[pb] = p.Bt + (long long)(b_ok[pb] ? n : 0) * p.b_ld + kq * 4;
    }

    // k-tile state machine for the fast path: k tile kt = (cs, tap), tap = (ky, kx)
    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;

    f32x4 ra[PA], rb[PB];

    auto gload = [&](int kt) {
        const int k0 = kt * 32;
        if constexpr (!GENERIC) {
            const int c0 = cs * 32 + kq * 4;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                int iy = a_iy0[pa] + ky;
                int ix = a_ix0[pa] + kx;
                const bool ok = a_ok[pa] && ((unsigned)iy < (unsigned)Hin) && ((unsigned)ix < (unsigned)Win);
                iy >>= p.ups;
                ix >>= p.ups;
                const long long off = (((long long)a_nb[pa] * p.Hs + iy) * p.Ws + ix) * p.a_ld + c0;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(p.A + off);
                ra[pa] = v;
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_ok[pb]) v = *reinterpret_cast<const f32x4*>(b_ptr[pb] + k0);
                rb[pb] = v;
            }
            // advance (kx, ky, cs)
            if (++kx == p.KW) {
                kx = 0;
                if (++ky == p.KH) { ky = 0; ++cs; }
            }
        } else {
            // generic path (Cin < 32, e.g. the 4-channel latent): decode each float4
            const int k = k0 + kq * 4;
            const bool kok = k < p.K;
            const int kk = kok ? k : 0;
            const int sl = kk / p.CS;          // slice*T + tap
            const int ci = kk - sl * p.CS;
            const int gcs = sl / T;
            const int gtap = sl - gcs * T;
            const int gky = gtap / p.KW;
            const int gkx = gtap - gky * p.KW;
            const int c0 = gcs * p.CS + ci;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                int iy = a_iy0[pa] + gky;
                int ix = a_ix0[pa] + gkx;
                const bool ok = kok && a_ok[pa] && ((unsigned)iy < (unsigned)Hin) && ((unsigned)ix < (unsigned)Win);
                iy >>= p.ups;
                ix >>= p.ups;
                const long long off = (((long long)a_nb[pa] * p.Hs + iy) * p.Ws + ix) * p.a_ld + c0;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(p.A + off);
                ra[pa] = v;
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kok && b_ok[pb]) v = *reinterpret_cast<const f32x4*>(b_ptr[pb] + k0);
                rb[pb] = v;
            }
        }
    };

    auto lstore = [&](int buf) {
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int r = pa * 32 + lrow;
            if (r < BM) *reinterpret_cast<f32x4*>(As + (buf * BM + r) * LD + kq * 4) = ra[pa];
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int r = pb * 32 + lrow;
            if (r < BN) *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * LD + kq * 4) = rb[pb];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag_off = (lane & 15) * LD + (lane >> 4) * 4;
    const float* a_frag_base = As + (wm * 16 * MI) * LD + frag_off;
    const float* b_frag_base = Bs + (wn * 16 * NI) * LD + frag_off;

    gload(kt_begin);
    lstore(0);
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = (kt + 1) < kt_end;
        if (more) gload(kt + 1);

        const float* Ab = a_frag_base + cur * BM * LD;
        const float* Bb = b_frag_base + cur * BN * LD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(Ab + mi * 16 * LD + kk * 16);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Bb + ni * 16 * LD + kk * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni][j], a[mi][j], acc[mi][ni], 0, 0, 0);
        }

        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------
    const bool split = p.splits > 1;
    float* Cbase = split ? (p.C + (long long)z * p.slab_stride) : p.C;
    const int ldc = split ? p.N : p.ldc;
    const bool vec_ok = ((p.N & 3) == 0) && ((ldc & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int smp = m / HoWo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[mi][ni];
            if (vec_ok) {
                if (!split) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                }
                *reinterpret_cast<f32x4*>(Cbase + (long long)m * ldc + n) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < p.N) {
                        float s = v[r];
                        if (!split) {
                            if (p.bias) s += p.bias[n + r];
                            if (p.rowvec) s += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                            if (p.resid) s += p.resid[(long long)m * p.ldr + n + r];
                        }
                        Cbase[(long long)m * ldc + n + r] = s;
                    }
                }
            }
        }
    }
}

// ---- split-K reduction + epilogue ------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvGemm p, const float* slabs, float* C) {
    const int HoWo = p.Ho * p.Wo;
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0);
    if (vec_ok) {
        const int n4 = p.N >> 2;
        const long long total = (long long)p.M * n4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int m = (int)(i / n4);
            const int n = (int)(i - (long long)m * n4) * 4;
            const long long off = (long long)m * p.N + n;
            f32x4 v = *reinterpret_cast<const f32x4*>(slabs + off);
            for (int s = 1; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(slabs + s * p.slab_stride + off);
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)(m / HoWo) * p.rowvec_stride + n);
            if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
            *reinterpret_cast<f32x4*>(C + (long long)m * p.ldc + n) = v;
        }
    } else {
        const long long total = (long long)p.M * p.N;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int m = (int)(i / p.N);
            const int n = (int)(i - (long long)m * p.N);
            float v = slabs[i];
            for (int s = 1; s < p.splits; ++s) v += slabs[s * p.slab_stride + i];
            if (p.bias) v += p.bias[n];
            if (p.rowvec) v += p.rowvec[(long long)(m / HoWo) * p.rowvec_stride + n];
            if (p.resid) v += p.resid[(long long)m * p.ldr + n];
            C[(long long)m * p.ldc + n] = v;
        }
    }
}

// ---- weight packing ------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ bt, int cout, int cin,
                                        int kh, int kw) {
    const int T = kh * kw;
    const int CS = cin < 32 ? cin : 32;
    const long long K = (long long)cin * T;
    const long long total = (long long)cout * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K);
        const int k = (int)(i - (long long)n * K);
        const int sl = k / CS;
        const int ci = k - sl * CS;
        const int cs = sl / T;
        const int tap = sl - cs * T;
        const int c = cs * CS + ci;
        bt[i] = w[((long long)n * cin + c) * T + tap];
    }
}

__global__ void pack_linear_weight_kernel(const float* __restrict__ w, float* __restrict__ bt, int cin, int cout) {
    // w [cin][cout] -> bt [cout][cin], 32x32 tiles through LDS
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over cout, by over cin
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int ci = by + r, co = bx + tx;
        tile[r][tx] = (ci < cin && co < cout) ? w[(long long)ci * cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = bx + r, ci = by + tx;
        if (co < cout && ci < cin) bt[(long long)co * cin + ci] = tile[tx][r];
    }
}

// ---- host side ----------------------------------------------------------------------
static const GemmTileInfo kTiles[kNumGemmTiles] = {
    {128, 128, "128x128"},  // 0: wave 64x64
    {128, 64, "128x64"},    // 1: wave 64x32
    {64, 64, "64x64"},      // 2: wave 32x32
    {256, 128, "256x128"},  // 3: wave 128x64
    {128, 80, "128x80"},    // 4: wave 32x80  (N = 320 -> 4 column tiles)
    {256, 80, "256x80"},    // 5: wave 64x80
    {64, 128, "64x128"},    // 6: wave 32x64
    {128, 160, "128x160"},  // 7: wave 64x80 (2x2 waves)
    {64, 80, "64x80"},      // 8: wave 16x80  (M = 8192, N = 320 -> 512 tiles, no split-K)
    {64, 160, "64x160"},    // 9: wave 32x80 (2x2 waves)
};

const GemmTileInfo& gemm_tile_info(int cfg) { return kTiles[cfg]; }

size_t gemm_tile_lds_bytes(int cfg) {
    return (size_t)2 * (kTiles[cfg].bm + kTiles[cfg].bn) * kLdsLd * sizeof(float);
}

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg(const ConvGemm& p, size_t lds, dim3 grid, hipStream_t stream) {
    const bool generic = (p.Cin % 32) != 0;
    if (generic) {
        auto k = conv_gemm_kernel<MI, NI, WM, WN, true>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    } else {
        auto k = conv_gemm_kernel<MI, NI, WM, WN, false>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_conv_gemm(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTiles) return hipErrorInvalidValue;
    const int bm = kTiles[cfg].bm, bn = kTiles[cfg].bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bn - 1) / bn;
    const int tiles = MT * NT;
    dim3 grid(((tiles + 7) / 8) * 8, 1, p.splits);
    const size_t lds = gemm_tile_lds_bytes(cfg);
    switch (cfg) {
        case 0: return launch_cfg<4, 4, 2, 2>(p, lds, grid, stream);
        case 1: return launch_cfg<4, 2, 2, 2>(p, lds, grid, stream);
        case 2: return launch_cfg<2, 2, 2, 2>(p, lds, grid, stream);
        case 3: return launch_cfg<8, 4, 2, 2>(p, lds, grid, stream);
        case 4: return launch_cfg<2, 5, 4, 1>(p, lds, grid, stream);
        case 5: return launch_cfg<4, 5, 4, 1>(p, lds, grid, stream);
        case 6: return launch_cfg<2, 4, 2, 2>(p, lds, grid, stream);
        case 7: return launch_cfg<4, 5, 2, 2>(p, lds, grid, stream);
        case 8: return launch_cfg<1, 5, 4, 1>(p, lds, grid, stream);
        case 9: return launch_cfg<2, 5, 2, 2>(p, lds, grid, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_splitk_reduce(const ConvGemm& p, const float* slabs, float* C, hipStream_t stream) {
    const long long work = ((long long)p.M * p.N + 3) / 4;
    int blocks = (int)((work + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p, slabs, C);
    return hipGetLastError();
}

hipError_t launch_pack_conv_weight(const float* w, float* bt, int cout, int cin, int kh, int kw, hipStream_t s) {
    const long long total = (long long)cout * cin * kh * kw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, s, w, bt, cout, cin, kh, kw);
    return hipGetLastError();
}

hipError_t launch_pack_linear_weight(const float* w, float* bt, int cin, int cout, hipStream_t s) {
    dim3 grid((cout + 31) / 32, (cin + 31) / 32);
    hipLaunchKernelGGL(pack_linear_weight_kernel, grid, dim3(256), 0, s, w, bt, cin, cout);
    return hipGetLastError();
}

}  // namespace sdmi
