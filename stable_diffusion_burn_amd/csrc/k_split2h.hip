// k_split2h.hip -- STAGED (not on any default path): the operand side of the two-term fp16 form of the fp32 GEMM.
//
// An fp32 value x, multiplied by a power of two s chosen so that the tensor's (or weight row's) largest magnitude lands in (2^13, 2^14],
// is stored as  h = fp16(s x)  and  l = fp16(s x - h)  (both round-to-nearest-even; s x - h is exact in fp32: h keeps the top 11 bits).
// h + l carries 22 significant bits of s x wherever |s x| >= 2^-3 and an absolute error <= 2^-25 below that (l subnormal): relative to
// the tensor's maximum that is <= 2^-38.  An fp16 x fp16 product is exact in fp32, the matrix instruction accumulates in fp32, and the
// GEMM adds the three partial products wl ah, wh al, wh ah (smallest first); the dropped wl al is <= 2^-22 of the product.  The scales are
// powers of two and factor out of the GEMM exactly: the epilogue multiplies the accumulators by their reciprocals (k_gemm_epi.hpp SCALED).
// Split error of the whole form measured on the host (tools/study_fp16_split.py): 0.7-1.8e-7 of max|C| for K = 320 ... 23040 -- below the
// fp32 summation error of any fp32 GEMM at those sizes -- at half the matrix instructions and two thirds of the operand bytes of the
// six-product bf16 form.  Layout of the planes: [row][C / 32][plane h, l][32] fp16 = 128 bytes per 32-channel slice, a plane row in the
// matrix lanes' order (chunk g = slice elements 4g..4g+3, 16+4g..16+4g+3: s3_plane_pos), i.e. k_gemm3p.hip's layout with two planes.
//
// Domain: finite inputs.  (+-inf / NaN make the scale meaningless; the staged operator-level path checks the tensor's maximum and refuses.)
#include "kernels.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s2h_split1(float xs, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)(xs - (float)hh);
    h = (unsigned)__builtin_bit_cast(unsigned short, hh);
    l = (unsigned)__builtin_bit_cast(unsigned short, ll);
}
// eight scaled values (chunk g of a slice: lo = elements 4g.., hi = elements 16+4g..) -> the chunk's 16 bytes of plane h and of plane l
__device__ __forceinline__ void s2h_split8(const f32x4 lo, const f32x4 hi, const float s, u32x4& ph, u32x4& pl) {
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { s2h_split1(lo[e] * s, h[e], l[e]); s2h_split1(hi[e] * s, h[4 + e], l[4 + e]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { ph[e] = h[2 * e] | (h[2 * e + 1] << 16); pl[e] = l[2 * e] | (l[2 * e + 1] << 16); }
}
// the power of two that moves a maximum magnitude of bit pattern `bits` (a non-negative finite float) into (2^13, 2^14]; 1 for a zero tensor
__device__ __forceinline__ float s2h_scale_of(unsigned bits, float* inv) {
    int e = (int)((bits >> 23) & 0xFFu) - 127;                       // floor(log2(max)) for normal numbers
    if (bits & 0x7FFFFFu) e += 1;                                     // ceil
    if ((bits >> 23) == 0u) e = bits ? -126 : 14;                     // subnormal maximum: treat as 2^-126; zero tensor: scale 1
    int se = 14 - e;                                                  // scale = 2^se
    se = se > 126 ? 126 : (se < -126 ? -126 : se);
    *inv = __builtin_bit_cast(float, (unsigned)(127 - se) << 23);
    return __builtin_bit_cast(float, (unsigned)(127 + se) << 23);
}

// ---- activations: tensor maximum -> scale -> planes ------------------------------------------------------------------------------
// amax_bits must be zero before the launch; non-negative floats order like their bit patterns, so atomicMax on the bits is a float max.
__global__ void absmax_bits_kernel(const float* __restrict__ x, long long rows, int c, long long ld, unsigned* __restrict__ amax_bits) {
    unsigned m = 0u;
    const long long total = rows * (c / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (c / 4);
        const int q = (int)(i - row * (c / 4));
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ld + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned b = __builtin_bit_cast(unsigned, v[e]) & 0x7FFFFFFFu; m = b > m ? b : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax_bits, m);
}
// scales[0] = s, scales[1] = 1 / s (what the GEMM epilogue multiplies by: ConvGemm::a_scale points at scales + 1)
__global__ void scale2h_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scales) {
    float inv;
    const float s = s2h_scale_of(*amax_bits, &inv);
    scales[0] = s;
    scales[1] = inv;
}
// x [rows][ld] fp32 (C = 32 kt channels used) -> y2 [rows][ld2 bytes / 128 slices][2][32] fp16.  One thread per (row, slice, chunk).
__global__ void split2h_rows_kernel(const float* __restrict__ x, unsigned short* __restrict__ y2, long long rows, int kt, long long ld, long long ld2_elems,
                                    const float* __restrict__ scales) {
    const float s = scales[0];
    const long long total = rows * kt * 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i & 3);
        const long long rk = i >> 2;
        const long long row = rk / kt;
        const int k = (int)(rk - row * kt);
        const float* src = x + row * ld + k * 32;
        u32x4 ph, pl;
        s2h_split8(*reinterpret_cast<const f32x4*>(src + 4 * g), *reinterpret_cast<const f32x4*>(src + 16 + 4 * g), s, ph, pl);
        unsigned short* dst = y2 + row * ld2_elems + k * 64 + g * 8;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + 32) = pl;
    }
}

// ---- weights: one workgroup per row: row maximum -> the row's scale -> planes + inv_scale[row] -------------------------------------
__global__ __launch_bounds__(256) void pack_split2h_kernel(const float* __restrict__ bt, unsigned short* __restrict__ w2, float* __restrict__ inv_scale, int K) {
    __shared__ unsigned s_max[4];
    const long long row = blockIdx.x;
    const float* src = bt + row * (long long)K;
    unsigned m = 0u;
    for (int q = threadIdx.x; q < K / 4; q += blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned b = __builtin_bit_cast(unsigned, v[e]) & 0x7FFFFFFFu; m = b > m ? b : m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    m = s_max[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) m = s_max[i] > m ? s_max[i] : m;
    float inv;
    const float s = s2h_scale_of(m, &inv);
    if (threadIdx.x == 0) inv_scale[row] = inv;
    const int kt = K / 32;
    for (int i = threadIdx.x; i < kt * 4; i += blockDim.x) {
        const int g = i & 3, k = i >> 2;
        u32x4 ph, pl;
        s2h_split8(*reinterpret_cast<const f32x4*>(src + k * 32 + 4 * g), *reinterpret_cast<const f32x4*>(src + k * 32 + 16 + 4 * g), s, ph, pl);
        unsigned short* dst = w2 + row * (long long)kt * 64 + k * 64 + g * 8;
        *reinterpret_cast<u32x4*>(dst) = ph;
        *reinterpret_cast<u32x4*>(dst + 32) = pl;
    }
}

hipError_t launch_absmax_bits(const float* x, long long rows, int c, long long ld, unsigned* amax_bits, hipStream_t s) {
    if ((c % 4) || (ld % 4)) return hipErrorInvalidValue;
    if (hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned), s); e != hipSuccess) return e;
    long long blocks = (rows * (c / 4) + 255) / 256;
    blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, rows, c, ld, amax_bits);
    return hipGetLastError();
}
hipError_t launch_scale2h(const unsigned* amax_bits, float* scales, hipStream_t s) {
    hipLaunchKernelGGL(scale2h_kernel, dim3(1), dim3(1), 0, s, amax_bits, scales);
    return hipGetLastError();
}
hipError_t launch_split2h_rows(const float* x, void* y2, long long rows, int c, long long ld, long long ld2_bytes, const float* scales, hipStream_t s) {
    if ((c % 32) || (ld % 4) || (ld2_bytes % 128) || ld2_bytes < (long long)(c / 32) * 128) return hipErrorInvalidValue;
    long long blocks = (rows * (c / 32) * 4 + 255) / 256;
    blocks = blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(split2h_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, reinterpret_cast<unsigned short*>(y2), rows, c / 32, ld, ld2_bytes / 2, scales);
    return hipGetLastError();
}
hipError_t launch_pack_split2h(const float* bt, void* w2, float* inv_scale, long long rows, int K, hipStream_t s) {
    if ((K % 32) || rows <= 0 || rows > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_split2h_kernel, dim3((unsigned)rows), dim3(256), 0, s, bt, reinterpret_cast<unsigned short*>(w2), inv_scale, K);
    return hipGetLastError();
}

}  // namespace sdmi
