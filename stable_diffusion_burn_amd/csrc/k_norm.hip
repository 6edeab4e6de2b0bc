// k_norm.hip -- the HBM-bound normalisation class: fused GroupNorm(+SiLU) over
// NHWC and row LayerNorm.
//
// Replaces GroupNorm::forward + layernorm (reference src/model/groupnorm/mod.rs:
// 53-82: reshape -> mean -> sub -> square-mean -> +eps -> sqrt -> div -> *gamma
// -> +beta, ~10 full tensor passes on the reference's backends) followed by
// SILU::forward (src/model/silu.rs:14-16, 2 more passes), and Burn's
// nn::LayerNorm (src/model/unet/mod.rs:523-525).
//
// MI355X mapping
//  * NHWC: a pixel's C channels are contiguous, so a workgroup of cq*R threads
//    (cq = C/4 float4 columns, R pixel rows per pass) streams whole pixel rows
//    with 16-byte loads, every wave instruction touching 1 KiB of consecutive
//    bytes.  Each thread owns one fixed float4 column, so its 4 channels'
//    gamma/beta/mean/rstd live in registers for the whole kernel.
//  * stats pass: the reference is two-pass (u = x - mean; mean(u^2), groupnorm/mod.rs:75-82).  Here every
//    thread accumulates SHIFTED sums  sum(x - p), sum((x - p)^2)  in fp32 over <= a few dozen pixels, with
//    the pivot p = the chunk's first pixel of the same channel (a sample of the data, so |x - p| is O(sigma)
//    and nothing cancels however large |mean| / sigma is).  Per channel the R thread rows are summed in fp64
//    in a FIXED order through LDS and turned into (mean_c, M2_c = sum (x - mean_c)^2); channels of a group
//    and later the chunks are merged with the pairwise (Chan) update written as one shifted pass in fp64.
//    No atomics: bit-reproducible.  The apply pass finishes the statistics in fp64 and subtracts the mean as
//    a float-float pair (hi + lo), so  x - mean  is as exact as the reference's fp64-free u = x - mean can be.
//    Algorithmic traffic: 2 reads + 1 write of the tensor; the second read of
//    UNet-sized tensors (<= 21 MB) is served from L2 / Infinity Cache.
//  * LayerNorm: 16 / 32 / 64 lanes per token row (4 / 2 / 1 rows per wave), the row (<= 2048 channels) is held in
//    registers, exact two-pass mean / variance with xor-shuffle reductions.
#include <algorithm>

#include "kernels.hpp"
#include "k_common.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Geometry: a block is cq*R threads (cq = C/4 float4 columns, R pixel rows per pass, <= 1024
// threads); a sample's hw rows are cut into `chunks` contiguous ranges of ~64 KiB, one block each,
// the same cut for the stats and the apply pass.
struct GnGeom { int cq, R, threads, chunks, rows_per_chunk; };

// min_wgs > 0 (round 5, option gn32_min_wgs): at least that many workgroups over the n samples, where the tensor has the rows -- a batch-1 tensor cut by size alone
// (80 chunks x 2 samples) leaves 96 CUs without a workgroup, and the apply pass is bound by the busy CUs' store path
// bytes per chunk of the STATISTICS pass: larger than the apply pass's 64 KB -- every apply workgroup merges all chunk partials of its sample, so fewer partials
// shorten the latency chain in front of its first store (profiles/r05o, r05p: 64 -> 128 KB, class -1.4 %).  A constant since round 6 (round 5's process-wide probe
// switch could select a cut finer than gn_partials_bytes() sizes the buffer for).
constexpr int kGn32StatsChunkKb = 128;
static inline GnGeom gn_geom(int hw, int c, int n = 1, int min_wgs = 0, int chunk_kb = 64) {
    GnGeom g;
    g.cq = c / 4;
    g.R = g.cq >= 1024 ? 1 : 1024 / g.cq;
    if (g.R > 32) g.R = 32;
    if (g.R > hw) g.R = hw;
    g.threads = g.cq * g.R;
    const long long bytes = (long long)hw * c * 4;
    long long chunks = (bytes + chunk_kb * 1024LL - 1) / (chunk_kb * 1024LL);
    if (min_wgs > 0 && chunks * n < min_wgs) chunks = (min_wgs + n - 1) / n;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    int rpc = (int)((hw + chunks - 1) / chunks);
    rpc = (rpc + g.R - 1) / g.R * g.R;
    g.rows_per_chunk = rpc;
    g.chunks = (hw + rpc - 1) / rpc;
    return g;
}

size_t gn_partials_bytes(int n, int hw, int c, int min_wgs) {
    const int min_apply = min_wgs & 0xFFFF, min_stats = (min_wgs >> 16) ? (min_wgs >> 16) - 1 : min_apply;
    return (size_t)n * gn_geom(hw, c, n, min_stats, kGn32StatsChunkKb).chunks * 64 * 2 * sizeof(double);   // the cut launch_group_norm_any makes
}

// Partial statistics of one (sample, chunk, group): part[((smp*chunks + chunk)*G + g)*2 + {0: mean, 1: M2}]
// over the chunk's rows x (C/G) channels, M2 = sum (x - mean)^2.  `ldx` = floats between pixels of x (>= C: the
// tensor may be a channel slice of a wider buffer).
__global__ void gn_stats_kernel(const float* __restrict__ x, int hw, int C, int ldx, int G, int rows_per_chunk,
                                double* __restrict__ part) {
    extern __shared__ float sh[];  // [2][R][C] floats, [C] pivots, then [2][C] doubles
    const int cq = C >> 2;
    const int R = blockDim.x / cq;
    const int tid = threadIdx.x;
    const int c4 = tid % cq;
    const int r0 = tid / cq;
    const int chunk = blockIdx.x, chunks = gridDim.x, smp = blockIdx.y;
    const int row_begin = chunk * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, hw);

    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, q0 = s0, s1 = s0, q1 = s0;
    const float* xb = x + (long long)smp * hw * ldx + c4 * 4;
    const f32x4 pv = *reinterpret_cast<const f32x4*>(xb + (long long)row_begin * ldx);  // pivot: same address for the R threads of a column
    int row = row_begin + r0;
    for (; row + R < row_end; row += 2 * R) {  // two independent loads in flight per thread
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (long long)row * ldx) - pv;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(xb + (long long)(row + R) * ldx) - pv;
        s0 += v0; q0 += v0 * v0;
        s1 += v1; q1 += v1 * v1;
    }
    if (row < row_end) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (long long)row * ldx) - pv;
        s0 += v0; q0 += v0 * v0;
    }
    s0 += s1; q0 += q1;
    float* shs = sh;
    float* shq = sh + R * C;
    float* shp = sh + 2 * R * C;
    double* chm = reinterpret_cast<double*>(sh + 2 * R * C + C);
    double* chq = chm + C;
    *reinterpret_cast<f32x4*>(shs + r0 * C + c4 * 4) = s0;
    *reinterpret_cast<f32x4*>(shq + r0 * C + c4 * 4) = q0;
    if (r0 == 0) *reinterpret_cast<f32x4*>(shp + c4 * 4) = pv;
    __syncthreads();
    const double n_rows = (double)(row_end - row_begin);
    const double inv_rows = 1.0 / n_rows;   // one fp64 division per thread: the per-channel / per-group ones below are multiplies
    for (int ch = tid; ch < C; ch += blockDim.x) {  // per channel: the R thread rows in fixed order -> (mean_c, M2_c)
        double ds = 0.0, dq = 0.0;
        for (int r = 0; r < R; ++r) { ds += (double)shs[r * C + ch]; dq += (double)shq[r * C + ch]; }
        const double m2 = dq - ds * ds * inv_rows;
        chm[ch] = (double)shp[ch] + ds * inv_rows;
        chq[ch] = m2 > 0.0 ? m2 : 0.0;
    }
    __syncthreads();
    gn_merge_group_channels(chm, chq, C, G, n_rows, part + (long long)(smp * chunks + chunk) * G * 2);
}

// P3: y is written as three bf16 planes (y points at [n][hw][C / 32][3][32] bf16) instead of fp32 -- the form the consuming
// k_gemm3p.hip launch reads (k_split3.hpp)
template <bool SILU, bool P3>
__global__ void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int hw, int C, int ldx, int G, float eps, int stat_chunks,
                                int stat_rows, const double* __restrict__ part, int rows_per_chunk) {
    __shared__ float s_mean_hi[64], s_mean_lo[64], s_rstd[64];
    __shared__ double s_red[3][64][8];
    const int cq = C >> 2;
    const int R = blockDim.x / cq;
    const int tid = threadIdx.x;
    const int smp = blockIdx.y;
    const int cpg = C / G;
    const int c4 = tid % cq;
    const int r0 = tid / cq;
    const int row_begin = blockIdx.x * rows_per_chunk;
    const int row_end = min(row_begin + rows_per_chunk, hw);
    const long long xbase = (long long)smp * hw * ldx + c4 * 4;
    const long long ybase = (long long)smp * hw * C + c4 * 4;
    // the thread's first two rows are requested BEFORE the statistics are finalised (round 5): gn_finalize is a chain of latencies (chunk partials from memory, fp64
    // merges, two barriers) during which nothing of the tensor was in flight
    int row = row_begin + r0;
    const bool head = row + R < row_end;
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
    if (head) {
        h0 = *reinterpret_cast<const f32x4*>(x + xbase + (long long)row * ldx);
        h1 = *reinterpret_cast<const f32x4*>(x + xbase + (long long)(row + R) * ldx);
    }
    // (round 6: gamma / beta too -- behind gn_finalize's barriers they were one more exposed round trip of a 7 - 9 us launch)
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
    gn_finalize(part, smp, G, cpg, hw, stat_chunks, stat_rows, eps, s_red, s_mean_hi, s_mean_lo, s_rstd);
    f32x4 mean_hi, mean_lo, rstd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gi = (c4 * 4 + i) / cpg;
        mean_hi[i] = s_mean_hi[gi];
        mean_lo[i] = s_mean_lo[gi];
        rstd[i] = s_rstd[gi];
    }
    auto norm = [&](f32x4 v) {
        v = ((v - mean_hi) - mean_lo) * rstd;
        v = v * gm + bt;
        if (SILU) {
#pragma unroll
            // x sigmoid(x) (silu.rs:14-16) with the hardware reciprocal (1 ulp) instead of the IEEE division's ten-instruction sequence: 88 of the unrolled body's
            // instructions in a latency-bound launch; the result moves by <= 2 ulp, far inside every bar
            for (int i = 0; i < 4; ++i) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[i]));
        }
        return v;
    };
    const long long pix3 = (long long)(C >> 5) * 192;   // bytes of one pixel's planes
    unsigned char* y3 = reinterpret_cast<unsigned char*>(y) + (long long)smp * hw * pix3;
    auto put = [&](int r, const f32x4 v) {
        if constexpr (P3) s3_store4(y3 + (long long)r * pix3, c4 * 4, v);
        else *reinterpret_cast<f32x4*>(y + ybase + (long long)r * C) = v;
    };
    if (head) {
        put(row, norm(h0));
        put(row + R, norm(h1));
        row += 2 * R;
    }
    for (; row + R < row_end; row += 2 * R) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + xbase + (long long)row * ldx);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + xbase + (long long)(row + R) * ldx);
        put(row, norm(v0));
        put(row + R, norm(v1));
    }
    if (row < row_end) put(row, norm(*reinterpret_cast<const f32x4*>(x + xbase + (long long)row * ldx)));
}

static hipError_t launch_group_norm_any(const float* x, void* y, bool planes, const float* gamma, const float* beta, int n, int hw, int c,
                                        int ldx, int n_group, float eps, bool silu, void* partials, hipStream_t stream, int min_wgs) {
    if ((c & 3) || (ldx & 3) || ldx < c || n_group > 64 || c % n_group) return hipErrorInvalidValue;
    if (c / 4 > 1024 || (planes && (c & 31))) return hipErrorInvalidValue;
    // min_wgs: low 16 bits = the apply pass's minimum workgroup count, bits 16.. = the statistics pass's (0 = the same): the apply pass re-reads every chunk partial of
    // its sample in every workgroup, so the two passes need not be cut alike
    const int min_apply = min_wgs & 0xFFFF, min_stats = (min_wgs >> 16) ? (min_wgs >> 16) - 1 : min_apply;
    const GnGeom gs = gn_geom(hw, c, n, min_stats, kGn32StatsChunkKb), g = gn_geom(hw, c, n, min_apply);
    double* part = reinterpret_cast<double*>(partials);
    const size_t lds = (size_t)(2 * gs.R + 1) * c * sizeof(float) + (size_t)2 * c * sizeof(double);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(gs.chunks, n), dim3(gs.threads), lds, stream, x, hw, c, ldx, n_group,
                       gs.rows_per_chunk, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    float* yf = reinterpret_cast<float*>(y);
#define SDMI_GN_APPLY(S, P)                                                                                                        \
    hipLaunchKernelGGL((gn_apply_kernel<S, P>), dim3(g.chunks, n), dim3(g.threads), 0, stream, x, yf, gamma, beta, hw, c, ldx, n_group, \
                       eps, gs.chunks, gs.rows_per_chunk, part, g.rows_per_chunk)
    if (planes) { if (silu) SDMI_GN_APPLY(true, true); else SDMI_GN_APPLY(false, true); }
    else { if (silu) SDMI_GN_APPLY(true, false); else SDMI_GN_APPLY(false, false); }
#undef SDMI_GN_APPLY
    return hipGetLastError();
}

hipError_t launch_group_norm(const float* x, float* y, const float* gamma, const float* beta, int n, int hw, int c, int ldx,
                             int n_group, float eps, bool silu, void* partials, hipStream_t stream, int min_wgs) {
    return launch_group_norm_any(x, y, false, gamma, beta, n, hw, c, ldx, n_group, eps, silu, partials, stream, min_wgs);
}
hipError_t launch_group_norm_planes(const float* x, void* y3, const float* gamma, const float* beta, int n, int hw, int c, int ldx,
                                    int n_group, float eps, bool silu, void* partials, hipStream_t stream, int min_wgs) {
    return launch_group_norm_any(x, y3, true, gamma, beta, n, hw, c, ldx, n_group, eps, silu, partials, stream, min_wgs);
}

// ---- LayerNorm: L lanes per row ---------------------------------------------------------------
// A wave handles 64 / L rows at once, L = 16 / 32 / 64 chosen so that a lane holds <= 8 float4 of its row: at C = 320 a
// one-wave-per-row kernel issues two loads of which the second keeps 16 lanes busy; with L = 16 every lane has 5 independent
// 16-byte loads in flight and the wave covers 4 rows (exact two-pass mean / variance as before, xor-shuffles inside the L lanes).
constexpr int kLnMaxVec = 8;  // float4 per lane

// NV > 0: the row is exactly NV float4 per lane (C = 4 L NV: 320 / 640 / 1280 at L = 16 / 32 / 64, NV = 5 -- every UNet width): no per-vector range tests.  The
// generic form (NV = 0) compiles all kLnMaxVec vectors with their predicates and zero fills: 1 230 - 1 870 instructions, 690 of them register moves, for a kernel whose
// launches are latency-bound at 8 - 10 us (round 5).  Same operations in the same order: bit-identical.
template <int L, bool P3, int NV = 0>
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int rows, int C, float eps) {
    constexpr int NVEC = NV > 0 ? NV : kLnMaxVec;
    constexpr int RPW = 64 / L;                       // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / L, l = lane % L;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
    const bool row_ok = row < rows;
    const int cq = C >> 2;
    const float* xr = x + (long long)(row_ok ? row : 0) * C;
    f32x4 v[NVEC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int f = l + i * L;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((NV > 0 || f < cq) && row_ok) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + f * 4);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    // round 6: gamma / beta are requested here, behind the row's loads and in front of the two reductions, not behind them -- the launch (6 - 8 us, 960 per
    // batch-1 image) is one memory round trip deep instead of two.  Exact-width instantiations only (40 registers); same operations: bit-identical.
    f32x4 gmv[NV > 0 ? NV : 1], btv[NV > 0 ? NV : 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            gmv[i] = *reinterpret_cast<const f32x4*>(gamma + (l + i * L) * 4);
            btv[i] = *reinterpret_cast<const f32x4*>(beta + (l + i * L) * 4);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int f = l + i * L;
        if (NV > 0 || f < cq) {
            const f32x4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    if (!row_ok) return;
    float* yr = y + (long long)row * C;
    unsigned char* yr3 = reinterpret_cast<unsigned char*>(y) + (long long)row * (C >> 5) * 192;   // P3: the row's planes
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int f = l + i * L;
        if (NV > 0 || f < cq) {
            const f32x4 gm = NV > 0 ? gmv[NV > 0 ? i : 0] : *reinterpret_cast<const f32x4*>(gamma + f * 4);
            const f32x4 bt = NV > 0 ? btv[NV > 0 ? i : 0] : *reinterpret_cast<const f32x4*>(beta + f * 4);
            const f32x4 o = (v[i] - mean) * rstd * gm + bt;
            if constexpr (P3) s3_store4(yr3, f * 4, o);
            else *reinterpret_cast<f32x4*>(yr + f * 4) = o;
        }
    }
}

template <bool P3>
static hipError_t launch_layer_norm_any(const float* x, float* y, const float* gamma, const float* beta, int rows, int c, float eps, hipStream_t stream) {
    if ((c & 3) || c > kLnMaxVec * 256 || (P3 && (c & 31))) return hipErrorInvalidValue;
    const int cq = c >> 2;
    if (cq == 16 * 5) hipLaunchKernelGGL((layer_norm_kernel<16, P3, 5>), dim3((rows + 15) / 16), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    else if (cq == 32 * 5) hipLaunchKernelGGL((layer_norm_kernel<32, P3, 5>), dim3((rows + 7) / 8), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    else if (cq == 64 * 5) hipLaunchKernelGGL((layer_norm_kernel<64, P3, 5>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    else if (cq <= 16 * kLnMaxVec)
        hipLaunchKernelGGL((layer_norm_kernel<16, P3>), dim3((rows + 15) / 16), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    else if (cq <= 32 * kLnMaxVec)
        hipLaunchKernelGGL((layer_norm_kernel<32, P3>), dim3((rows + 7) / 8), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    else
        hipLaunchKernelGGL((layer_norm_kernel<64, P3>), dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, gamma, beta, rows, c, eps);
    return hipGetLastError();
}
hipError_t launch_layer_norm(const float* x, float* y, const float* gamma, const float* beta, int rows, int c,
                             float eps, hipStream_t stream) {
    return launch_layer_norm_any<false>(x, y, gamma, beta, rows, c, eps, stream);
}
hipError_t launch_layer_norm_planes(const float* x, void* y3, const float* gamma, const float* beta, int rows, int c, float eps,
                                    hipStream_t stream) {
    return launch_layer_norm_any<true>(x, reinterpret_cast<float*>(y3), gamma, beta, rows, c, eps, stream);
}

}  // namespace sdmi
