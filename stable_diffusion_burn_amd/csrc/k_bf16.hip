// k_bf16.hip -- bf16-storage variants of the HBM-bound kernels (precision = 1).
//
// Same algorithms as k_norm.hip / k_elem.hip (citations there); activations are bf16 in HBM, all
// arithmetic is fp32, statistics are combined in fp64 in fixed order.  A 16-byte access now carries
// 8 channels, so a thread owns one 8-channel column and the HBM traffic of every pass is halved.
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct F8 { float v[8]; };

__device__ __forceinline__ F8 unpack8(u32x4 w) {
    F8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.v[2 * i] = __uint_as_float(w[i] << 16);
        r.v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
    return r;
}
typedef float h_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 h_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ u32x4 pack8(const F8& f) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // one v_cvt_pk_bf16_f32 per pair (round to nearest even, as bf16_bits)
        const h_f32x2 p2 = {f.v[2 * i], f.v[2 * i + 1]};
        w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(p2, h_bf16x2));
    }
    return w;
}

static inline int blocks_for(long long work, int cap = 2048) {
    long long b = (work + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

#define GRID_STRIDE(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// ---- GroupNorm (+SiLU) ------------------------------------------------------------------------------
GnGeomH gn_geom_bf16(int n, int hw, int c, GnTune t) {
    GnGeomH g;
    const int maxt = t.max_threads >= 64 ? t.max_threads : 1024;
    g.cq = c / 8;
    g.R = g.cq >= maxt ? 1 : maxt / g.cq;
    if (g.R > 32) g.R = 32;
    if (g.R > hw) g.R = hw;
    g.threads = g.cq * g.R;
    const long long bytes = (long long)hw * c * 2;
    long long chunks = (bytes + 32767) / 32768;   // 32 KB of bf16 = as many rows (and workgroups) per chunk as the fp32 kernels' 64 KB
    if (t.target_wgs > 0) {                       // ... but not more workgroups than one round of the chip needs (kernels.hpp, GnTune)
        const long long per_sample = (t.target_wgs + (n > 0 ? n : 1) - 1) / (n > 0 ? n : 1);
        if (chunks > per_sample) chunks = per_sample;
    }
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    int rpc = (int)((hw + chunks - 1) / chunks);
    rpc = (rpc + g.R - 1) / g.R * g.R;
    g.rows_per_chunk = rpc;
    g.chunks = (hw + rpc - 1) / rpc;
    return g;
}

size_t gn_partials_bytes_bf16(int n, int hw, int c, GnTune t) { return (size_t)n * gn_geom_bf16(n, hw, c, t).chunks * 64 * 2 * sizeof(double); }

// shifted statistics, partial format and merge order: k_norm.hip / k_common.hpp.  `ldx` = elements between pixels of x.  U independent loads per thread are in
// flight; the values are consumed in row order, so a thread's sums do not depend on U.
template <int U>
__global__ void gn_stats_bf16_kernel(const unsigned short* __restrict__ x, int hw, int C, int ldx, int G, int rows_per_chunk,
                                     double* __restrict__ part) {
    extern __shared__ float sh[];  // [2][R][C] floats, [C] pivots, then [2][C] doubles
    const int cq = C >> 3;
    const int R = blockDim.x / cq;
    const int tid = threadIdx.x;
    const int c8 = tid % cq;
    const int r0 = tid / cq;
    const int chunk = blockIdx.x, chunks = gridDim.x, smp = blockIdx.y;
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, hw);
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    const unsigned short* xb = x + (long long)smp * hw * ldx + c8 * 8;
    const F8 pv = unpack8(*reinterpret_cast<const u32x4*>(xb + (long long)row_begin * ldx));
    auto add = [&](const u32x4 w) {
        const F8 v = unpack8(w);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v.v[i] - pv.v[i]; s[i] += d; q[i] += d * d; }
    };
    int row = row_begin + r0;
    if constexpr (U > 1) {
        for (; row + (U - 1) * R < row_end; row += U * R) {
            u32x4 w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const u32x4*>(xb + (long long)(row + u * R) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u) add(w[u]);
        }
    }
    for (; row < row_end; row += R) add(*reinterpret_cast<const u32x4*>(xb + (long long)row * ldx));
    float* shs = sh;
    float* shq = sh + R * C;
    float* shp = sh + 2 * R * C;
    double* chm = reinterpret_cast<double*>(sh + 2 * R * C + C);
    double* chq = chm + C;
#pragma unroll
    for (int i = 0; i < 8; ++i) { shs[r0 * C + c8 * 8 + i] = s[i]; shq[r0 * C + c8 * 8 + i] = q[i]; }
    if (r0 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) shp[c8 * 8 + i] = pv.v[i];
    }
    __syncthreads();
    const double n_rows = (double)(row_end - row_begin);
    const double inv_rows = 1.0 / n_rows;   // one fp64 division per thread: the per-channel / per-group ones below are multiplies
    for (int ch = tid; ch < C; ch += blockDim.x) {
        double ds = 0.0, dq = 0.0;
        for (int r = 0; r < R; ++r) { ds += (double)shs[r * C + ch]; dq += (double)shq[r * C + ch]; }
        const double m2 = dq - ds * ds * inv_rows;
        chm[ch] = (double)shp[ch] + ds * inv_rows;
        chq[ch] = m2 > 0.0 ? m2 : 0.0;
    }
    __syncthreads();
    gn_merge_group_channels(chm, chq, C, G, n_rows, part + (long long)(smp * chunks + chunk) * G * 2);
}

template <bool SILU, int U>
__global__ void gn_apply_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, int hw, int C, int ldx, int G,
                                     float eps, int stat_chunks, int stat_rows, const double* __restrict__ part, int rows_per_chunk) {
    __shared__ float s_mean_hi[64], s_mean_lo[64], s_rstd[64];
    __shared__ double s_red[3][64][8];
    const int cq = C >> 3;
    const int R = blockDim.x / cq;
    const int tid = threadIdx.x;
    const int smp = blockIdx.y;
    const int cpg = C / G;
    const int c8 = tid % cq;
    const int r0 = tid / cq;
    const int row_begin = blockIdx.x * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, hw);
    const long long xbase = (long long)smp * hw * ldx + c8 * 8;
    const long long ybase = (long long)smp * hw * C + c8 * 8;
    // the thread's first rows are requested BEFORE the statistics are finalised (round 5): gn_finalize is a chain of latencies -- every workgroup reads its sample's
    // chunk partials, merges them in fp64, crosses two barriers -- during which nothing of the tensor was in flight
    int row = row_begin + r0;
    u32x4 w0[U];
    const bool head = row + (U - 1) * R < row_end;
    if (head) {
#pragma unroll
        for (int u = 0; u < U; ++u) w0[u] = *reinterpret_cast<const u32x4*>(x + xbase + (long long)(row + u * R) * ldx);
    }
    // (round 6: gamma / beta too, as two 16-byte loads each -- behind gn_finalize's barriers they were one more exposed round trip per workgroup)
    float gr[8], bt[8], mean_hi[8], mean_lo[8];
    {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c8 * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + c8 * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c8 * 8), b1 = *reinterpret_cast<const f32x4*>(beta + c8 * 8 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gr[i] = g0[i]; gr[4 + i] = g1[i]; bt[i] = b0[i]; bt[4 + i] = b1[i]; }
    }
    gn_finalize(part, smp, G, cpg, hw, stat_chunks, stat_rows, eps, s_red, s_mean_hi, s_mean_lo, s_rstd);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ch = c8 * 8 + i;
        mean_hi[i] = s_mean_hi[ch / cpg];
        mean_lo[i] = s_mean_lo[ch / cpg];
    }
    float rstd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rstd[i] = s_rstd[(c8 * 8 + i) / cpg];
    auto norm = [&](const u32x4 w) {
        F8 v = unpack8(w);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = ((v.v[i] - mean_hi[i]) - mean_lo[i]) * rstd[i];
            t = t * gr[i] + bt[i];
            if (SILU) t = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));   // (hardware reciprocal, 1 ulp: at the batches of configs[2..4] the IEEE division made this pass VALU-bound)
            v.v[i] = t;
        }
        return pack8(v);
    };
    if (head) {
#pragma unroll
        for (int u = 0; u < U; ++u) *reinterpret_cast<u32x4*>(y + ybase + (long long)(row + u * R) * C) = norm(w0[u]);
        row += U * R;
    }
    if constexpr (U > 1) {
        for (; row + (U - 1) * R < row_end; row += U * R) {
            u32x4 w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const u32x4*>(x + xbase + (long long)(row + u * R) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u) *reinterpret_cast<u32x4*>(y + ybase + (long long)(row + u * R) * C) = norm(w[u]);
        }
    }
    for (; row < row_end; row += R) *reinterpret_cast<u32x4*>(y + ybase + (long long)row * C) = norm(*reinterpret_cast<const u32x4*>(x + xbase + (long long)row * ldx));
}

hipError_t launch_group_norm_bf16_stats(const void* x, int n, int hw, int c, int ldx, int n_group, void* partials, hipStream_t stream, GnTune t) {
    if ((c & 7) || (ldx & 7) || ldx < c || n_group > 64 || c % n_group || c / 8 > 1024) return hipErrorInvalidValue;
    const GnGeomH g = gn_geom_bf16(n, hw, c, t);
    const size_t lds = (size_t)(2 * g.R + 1) * c * sizeof(float) + (size_t)2 * c * sizeof(double);
    auto xs = reinterpret_cast<const unsigned short*>(x);
    double* part = reinterpret_cast<double*>(partials);
    if (t.unroll >= 4) hipLaunchKernelGGL(gn_stats_bf16_kernel<4>, dim3(g.chunks, n), dim3(g.threads), lds, stream, xs, hw, c, ldx, n_group, g.rows_per_chunk, part);
    else if (t.unroll >= 2) hipLaunchKernelGGL(gn_stats_bf16_kernel<2>, dim3(g.chunks, n), dim3(g.threads), lds, stream, xs, hw, c, ldx, n_group, g.rows_per_chunk, part);
    else hipLaunchKernelGGL(gn_stats_bf16_kernel<1>, dim3(g.chunks, n), dim3(g.threads), lds, stream, xs, hw, c, ldx, n_group, g.rows_per_chunk, part);
    return hipGetLastError();
}

template <bool SILU>
static void launch_gn_apply_bf16(const GnGeomH& g, int unroll, int n, hipStream_t stream, const unsigned short* xs, unsigned short* ys, const float* gamma, const float* beta,
                                 int hw, int c, int ldx, int n_group, float eps, const double* part) {
    if (unroll >= 2) hipLaunchKernelGGL((gn_apply_bf16_kernel<SILU, 2>), dim3(g.chunks, n), dim3(g.threads), 0, stream, xs, ys, gamma, beta, hw, c, ldx, n_group, eps,
                                        g.chunks, g.rows_per_chunk, part, g.rows_per_chunk);
    else hipLaunchKernelGGL((gn_apply_bf16_kernel<SILU, 1>), dim3(g.chunks, n), dim3(g.threads), 0, stream, xs, ys, gamma, beta, hw, c, ldx, n_group, eps,
                            g.chunks, g.rows_per_chunk, part, g.rows_per_chunk);
}

hipError_t launch_group_norm_bf16(const void* x, void* y, const float* gamma, const float* beta, int n, int hw, int c, int ldx,
                                  int n_group, float eps, bool silu, void* partials, hipStream_t stream, GnTune t) {
    hipError_t e = launch_group_norm_bf16_stats(x, n, hw, c, ldx, n_group, partials, stream, t);
    if (e != hipSuccess) return e;
    const GnGeomH g = gn_geom_bf16(n, hw, c, t);
    const double* part = reinterpret_cast<const double*>(partials);
    auto xs = reinterpret_cast<const unsigned short*>(x);
    auto ys = reinterpret_cast<unsigned short*>(y);
    if (silu) launch_gn_apply_bf16<true>(g, t.unroll, n, stream, xs, ys, gamma, beta, hw, c, ldx, n_group, eps, part);
    else launch_gn_apply_bf16<false>(g, t.unroll, n, stream, xs, ys, gamma, beta, hw, c, ldx, n_group, eps, part);
    return hipGetLastError();
}

// ---- LayerNorm: L lanes per row, 8-channel vectors (structure and reasons: k_norm.hip) ---------------------------------
constexpr int kLnMaxVecH = 4;  // 16-byte vectors per lane

template <int L>
__global__ __launch_bounds__(256) void layer_norm_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int rows,
                                                              int C, float eps) {
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63;
    const int sub = lane / L, l = lane % L;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool row_ok = row < rows;
    const int cq = C >> 3;
    const unsigned short* xr = x + (long long)(row_ok ? row : 0) * C;
    F8 v[kLnMaxVecH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxVecH; ++i) {
        const int f = l + i * L;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i].v[j] = 0.f;
        if (f < cq && row_ok) {
            v[i] = unpack8(*reinterpret_cast<const u32x4*>(xr + f * 8));
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i].v[j];
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxVecH; ++i) {
        const int f = l + i * L;
        if (f < cq) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i].v[j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    if (!row_ok) return;
    unsigned short* yr = y + (long long)row * C;
#pragma unroll
    for (int i = 0; i < kLnMaxVecH; ++i) {
        const int f = l + i * L;
        if (f < cq) {
            F8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = (v[i].v[j] - mean) * rstd * gamma[f * 8 + j] + beta[f * 8 + j];
            *reinterpret_cast<u32x4*>(yr + f * 8) = pack8(o);
        }
    }
}

hipError_t launch_layer_norm_bf16(const void* x, void* y, const float* gamma, const float* beta, int rows, int c, float eps,
                                  hipStream_t stream) {
    if ((c & 7) || c > kLnMaxVecH * 512) return hipErrorInvalidValue;
    const int cq = c >> 3;
    auto xs = reinterpret_cast<const unsigned short*>(x);
    auto ys = reinterpret_cast<unsigned short*>(y);
    if (cq <= 16 * kLnMaxVecH)
        hipLaunchKernelGGL(layer_norm_bf16_kernel<16>, dim3((rows + 15) / 16), dim3(256), 0, stream, xs, ys, gamma, beta, rows, c, eps);
    else if (cq <= 32 * kLnMaxVecH)
        hipLaunchKernelGGL(layer_norm_bf16_kernel<32>, dim3((rows + 7) / 8), dim3(256), 0, stream, xs, ys, gamma, beta, rows, c, eps);
    else
        hipLaunchKernelGGL(layer_norm_bf16_kernel<64>, dim3((rows + 3) / 4), dim3(256), 0, stream, xs, ys, gamma, beta, rows, c, eps);
    return hipGetLastError();
}

// ---- elementwise ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf_h(float x) { return gelu_gate_fast(x); }   // the fused epilogue's form (k_common.hpp), so that fused and unfused gates agree

__global__ void geglu_bf16_kernel(const unsigned short* __restrict__ proj, unsigned short* __restrict__ out, long long rows, int hidden8) {
    const long long total = rows * hidden8;
    GRID_STRIDE(i, total) {
        const long long r = i / hidden8;
        const int c = (int)(i - r * hidden8);
        const F8 a = unpack8(reinterpret_cast<const u32x4*>(proj)[r * 2 * hidden8 + c]);
        const F8 g = unpack8(reinterpret_cast<const u32x4*>(proj)[r * 2 * hidden8 + hidden8 + c]);
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = a.v[j] * gelu_erf_h(g.v[j]);
        reinterpret_cast<u32x4*>(out)[i] = pack8(o);
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n) {
    GRID_STRIDE(i, n) dst[i] = (unsigned short)bf16_bits(src[i]);
}
__global__ void f32_to_bf16_scaled_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n, float scale) {
    GRID_STRIDE(i, n) dst[i] = (unsigned short)bf16_bits(src[i] * scale);
}
__global__ void scale_f32_kernel(float* __restrict__ x, long long n, float scale) {
    GRID_STRIDE(i, n) x[i] *= scale;
}

__global__ void nhwc_bf16_to_nchw_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int n, int c, int h, int w) {
    const long long hw = (long long)h * w, total = (long long)n * c * hw;
    GRID_STRIDE(i, total) {
        const long long p = i % hw;
        const long long bc = i / hw;
        const int ch = (int)(bc % c);
        const long long b = bc / c;
        dst[i] = __uint_as_float((unsigned)src[(b * hw + p) * c + ch] << 16);
    }
}

__global__ void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int n, int c, int h, int w,
                                             float scale) {
    const long long hw = (long long)h * w, total = (long long)n * c * hw;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % c);
        const long long px = i / c;
        const long long b = px / hw, p = px - b * hw;
        dst[i] = (unsigned short)bf16_bits(src[(b * c + ch) * hw + p] * scale);
    }
}

__global__ void transpose2d_bf16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols,
                                        int src_ld) {
    __shared__ unsigned short tile[32][34];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int rr = by + r, cc = bx + tx;
        tile[r][tx] = (rr < rows && cc < cols) ? src[(long long)rr * src_ld + cc] : (unsigned short)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int cc = bx + r, rr = by + tx;
        if (cc < cols && rr < rows) dst[(long long)cc * rows + rr] = tile[tx][r];
    }
}

// row softmax of the fp32 score matrix of the unfused single-head (VAE) attention, bf16 probabilities out
__global__ __launch_bounds__(256) void softmax_rows_f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                                       int rows, int cols, float scale) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    if (row >= rows) return;
    const float* xr = x + (long long)row * cols;
    unsigned short* yr = y + (long long)row * cols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, xr[i] * scale);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < cols; i += 256) sum += __expf(xr[i] * scale - mx);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / sum;
    for (int i = tid; i < cols; i += 256) yr[i] = (unsigned short)bf16_bits(__expf(xr[i] * scale - mx) * inv);
}

hipError_t launch_geglu_bf16(const void* proj, void* out, long long rows, int hidden, hipStream_t s) {
    if (hidden & 7) return hipErrorInvalidValue;
    const long long total = rows * (hidden / 8);
    hipLaunchKernelGGL(geglu_bf16_kernel, dim3(blocks_for(total, 4096)), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(proj),
                       reinterpret_cast<unsigned short*>(out), rows, hidden / 8);
    return hipGetLastError();
}
hipError_t launch_f32_to_bf16(const float* src, void* dst, long long n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(n)), dim3(256), 0, s, src, reinterpret_cast<unsigned short*>(dst), n);
    return hipGetLastError();
}
hipError_t launch_f32_to_bf16_scaled(const float* src, void* dst, long long n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_scaled_kernel, dim3(blocks_for(n)), dim3(256), 0, s, src, reinterpret_cast<unsigned short*>(dst), n, scale);
    return hipGetLastError();
}
hipError_t launch_scale_f32(float* x, long long n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, n, scale);
    return hipGetLastError();
}
hipError_t launch_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int n, int c, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(blocks_for((long long)n * c * h * w)), dim3(256), 0, s,
                       reinterpret_cast<const unsigned short*>(src), dst, n, c, h, w);
    return hipGetLastError();
}
hipError_t launch_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w, float scale, hipStream_t s) {
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3(blocks_for((long long)n * c * h * w)), dim3(256), 0, s, src,
                       reinterpret_cast<unsigned short*>(dst), n, c, h, w, scale);
    return hipGetLastError();
}
hipError_t launch_transpose2d_bf16(const void* src, void* dst, int rows, int cols, int src_ld, hipStream_t s) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(transpose2d_bf16_kernel, grid, dim3(256), 0, s, reinterpret_cast<const unsigned short*>(src),
                       reinterpret_cast<unsigned short*>(dst), rows, cols, src_ld);
    return hipGetLastError();
}
hipError_t launch_softmax_rows_f32_to_bf16(const float* x, void* y, int rows, int cols, float scale, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_f32_to_bf16_kernel, dim3(rows), dim3(256), 0, s, x, reinterpret_cast<unsigned short*>(y), rows, cols, scale);
    return hipGetLastError();
}

}  // namespace sdmi
