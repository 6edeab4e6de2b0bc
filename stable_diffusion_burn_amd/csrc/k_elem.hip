// k_elem.hip -- elementwise / data-movement kernels of the sampling path.
//
//  * layout conversion at the ABI boundary (reference tensors are NCHW,
//    src/model/stablediffusion/mod.rs:115-121; everything inside is NHWC)
//  * channel concat of the UNet skip connections (Tensor::cat(..,1), unet/mod.rs:134)
//  * GEGLU gate  a * gelu_erf(gate)  (unet/mod.rs:579-591, Burn Gelu = exact erf)
//  * SiLU of the time embedding (unet/mod.rs:117, 718)
//  * timestep_embedding (unet/mod.rs:19-30), f32 math like the reference
//  * CFG combine + DDIM update (stablediffusion/mod.rs:152-156, 190-191): the
//    reference's ~8 elementwise launches and 2 host syncs per step are one kernel
//  * image post-processing to u8 (stablediffusion/mod.rs:79-99)
// All are HBM/latency bound and tiny next to the convolutions; they use 16-byte
// accesses and grid-stride loops.
#include "kernels.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int blocks_for(long long work, int cap = 2048) {
    long long b = (work + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

#define GRID_STRIDE(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c, int h, int w,
                                    float scale) {
    const long long hw = (long long)h * w, total = (long long)n * c * hw;
    GRID_STRIDE(i, total) {  // i indexes dst (NHWC)
        const int ch = (int)(i % c);
        const long long px = i / c;
        const long long b = px / hw, p = px - b * hw;
        dst[i] = src[(b * c + ch) * hw + p] * scale;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c, int h,
                                    int w) {
    const long long hw = (long long)h * w, total = (long long)n * c * hw;
    GRID_STRIDE(i, total) {  // i indexes dst (NCHW)
        const long long p = i % hw;
        const long long bc = i / hw;
        const int ch = (int)(bc % c);
        const long long b = bc / c;
        dst[i] = src[(b * hw + p) * c + ch];
    }
}

// first c_out of c_src channels: NHWC [n,h,w,c_src] -> NCHW [n,c_out,h,w]
__global__ void nhwc_to_nchw_slice_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c_src, int c_out,
                                          long long hw) {
    const long long total = (long long)n * c_out * hw;
    GRID_STRIDE(i, total) {
        const long long p = i % hw;
        const long long bc = i / hw;
        const int ch = (int)(bc % c_out);
        const long long b = bc / c_out;
        dst[i] = src[(b * hw + p) * c_src + ch];
    }
}

__global__ void concat_channels_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       float* __restrict__ dst, long long rows, int ca4, int cb4) {
    const int ct4 = ca4 + cb4;
    const long long total = rows * ct4;
    GRID_STRIDE(i, total) {
        const long long r = i / ct4;
        const int c = (int)(i - r * ct4);
        f32x4 v;
        if (c < ca4) v = reinterpret_cast<const f32x4*>(a)[r * ca4 + c];
        else v = reinterpret_cast<const f32x4*>(b)[r * cb4 + (c - ca4)];
        reinterpret_cast<f32x4*>(dst)[i] = v;
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ void geglu_kernel(const float* __restrict__ proj, float* __restrict__ out, long long rows, int hidden4) {
    const long long total = rows * hidden4;
    GRID_STRIDE(i, total) {
        const long long r = i / hidden4;
        const int c = (int)(i - r * hidden4);
        const f32x4 a = reinterpret_cast<const f32x4*>(proj)[r * 2 * hidden4 + c];
        const f32x4 g = reinterpret_cast<const f32x4*>(proj)[r * 2 * hidden4 + hidden4 + c];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] * gelu_erf(g[j]);
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        y[i] = v * (1.0f / (1.0f + expf(-v)));
    }
}

// RGB image NCHW [n,3,h,w] -> NHWC with a zero 4th channel [n,h,w,4] (the VAE encoder's conv_in runs with Cin = 4)
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, long long hw) {
    GRID_STRIDE(i, (long long)n * hw) {
        const long long b = i / hw, p = i - b * hw;
        const float* s = src + b * 3 * hw + p;
        reinterpret_cast<float4*>(dst)[i] = make_float4(s[0], s[hw], s[2 * hw], 0.f);
    }
}

// QuickGELU (clip/mod.rs:223-225): x * sigmoid(1.702 x), in place
__global__ void quick_gelu_kernel(float* __restrict__ x, long long n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        x[i] = v * (1.0f / (1.0f + expf(-1.702f * v)));
    }
}

// CLIP::forward embedding (clip/mod.rs:62-67): out[b, t, :] = token_table[tokens[b, t], :] + position_table[t, :]
__global__ void clip_embed_kernel(const int* __restrict__ tokens, const float* __restrict__ tok_table,
                                  const float* __restrict__ pos_table, float* __restrict__ out, int n, int T, int C) {
    const long long total = (long long)n * T * (C / 4);
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % (C / 4));
        const long long row = i / (C / 4);
        const int t = (int)(row % T);
        const float4 a = reinterpret_cast<const float4*>(tok_table + (long long)tokens[row] * C)[c4];
        const float4 b = reinterpret_cast<const float4*>(pos_table + (long long)t * C)[c4];
        reinterpret_cast<float4*>(out + row * C)[c4] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// attn_decoder_mask (backend.rs:130-139): 0 on and below the diagonal, -inf above
__global__ void causal_mask_kernel(float* __restrict__ mask, int T) {
    GRID_STRIDE(i, (long long)T * T) {
        const int r = (int)(i / T), c = (int)(i % T);
        mask[i] = c > r ? -INFINITY : 0.f;
    }
}

__global__ void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols,
                                   int src_ld) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over cols, by over rows
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int rr = by + r, cc = bx + tx;
        tile[r][tx] = (rr < rows && cc < cols) ? src[(long long)rr * src_ld + cc] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int cc = bx + r, rr = by + tx;
        if (cc < cols && rr < rows) dst[(long long)cc * rows + rr] = tile[tx][r];
    }
}

__global__ void timestep_embedding_kernel(const int* __restrict__ t, int n_t, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    const float coef = (float)(-log(10000.0) / (double)half);  // f64 scalar narrowed to f32 (unet/mod.rs:25-27)
    const int total = n_t * half;
    GRID_STRIDE(i, total) {
        const int s = (int)(i / half), j = (int)(i - (long long)s * half);
        const float freq = expf((float)j * coef);
        const float arg = (float)t[s] * freq;
        out[(long long)s * dim + j] = cosf(arg);
        out[(long long)s * dim + half + j] = sinf(arg);
    }
}

__global__ void cfg_ddim_kernel(const float* __restrict__ eps, float* __restrict__ latent, float* __restrict__ unet_in,
                                long long per_half, DdimCoef c) {
    GRID_STRIDE(i, per_half) {
        const float eu = eps[i];
        const float ec = eps[per_half + i];
        const float e = eu + (ec - eu) * c.scale;                  // :190-191
        const float x = latent[i];
        const float predx0 = (x - e * c.sqrt_noise) / c.sqrt_cur;  // :152
        const float dir = e * c.dir_coef;                          // :153
        const float nx = predx0 * c.sqrt_prev + dir;               // :155 (sigma = 0)
        latent[i] = nx;
        unet_in[i] = nx;
        unet_in[per_half + i] = nx;
    }
}

__global__ void dup_latent_kernel(const float* __restrict__ latent, float* __restrict__ unet_in, long long per_half) {
    GRID_STRIDE(i, per_half) {
        const float x = latent[i];
        unet_in[i] = x;
        unet_in[per_half + i] = x;
    }
}

__global__ void image_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, long long n) {
    GRID_STRIDE(i, n) {
        float v = (img[i] + 1.0f) / 2.0f;  // :79
        v = v * 255.0f;                    // :84
        v = fminf(v, 255.0f);              // :96  .min(255).max(0) as u8 (truncation)
        v = fmaxf(v, 0.0f);
        out[i] = (uint8_t)v;
    }
}

// splitmix64 -> Box-Muller; used only when the caller passes no initial latent
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void fill_normal_kernel(float* __restrict__ dst, long long n, uint64_t seed) {
    GRID_STRIDE(i, n) {
        const uint64_t r = splitmix64(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
        const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
        const float u2 = (float)(uint32_t)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
        dst[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, float scale, hipStream_t s) {
    const long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, s, src, dst, n, c, h, w, scale);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, hipStream_t s) {
    const long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, s, src, dst, n, c, h, w);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw_slice(const float* src, float* dst, int n, int c_src, int c_out, int h, int w, hipStream_t s) {
    if (c_out > c_src) return hipErrorInvalidValue;
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_slice_kernel, dim3(blocks_for((long long)n * c_out * hw)), dim3(256), 0, s, src, dst, n, c_src, c_out, hw);
    return hipGetLastError();
}
hipError_t launch_concat_channels(const float* a, const float* b, float* dst, long long rows, int ca, int cb,
                                  hipStream_t s) {
    if ((ca & 3) || (cb & 3)) return hipErrorInvalidValue;
    const long long total = rows * ((ca + cb) / 4);
    hipLaunchKernelGGL(concat_channels_kernel, dim3(blocks_for(total)), dim3(256), 0, s, a, b, dst, rows, ca / 4,
                       cb / 4);
    return hipGetLastError();
}
// the same gate with the result as three bf16 planes ([rows][hidden / 32][3][32]; k_split3.hpp) for the k_gemm3p.hip launch of the MLP's second Linear
__global__ void geglu_planes_kernel(const float* __restrict__ proj, unsigned char* __restrict__ out3, long long rows, int hidden4) {
    const long long total = rows * hidden4;
    const long long row3 = (long long)(hidden4 >> 3) * 192;
    GRID_STRIDE(i, total) {
        const long long r = i / hidden4;
        const int c = (int)(i - r * hidden4);
        const f32x4 a = reinterpret_cast<const f32x4*>(proj)[r * 2 * hidden4 + c];
        const f32x4 g = reinterpret_cast<const f32x4*>(proj)[r * 2 * hidden4 + hidden4 + c];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] * gelu_erf(g[j]);
        s3_store4(out3 + r * row3, c * 4, o);
    }
}
hipError_t launch_geglu_planes(const float* proj, void* out3, long long rows, int hidden, hipStream_t s) {
    if (hidden & 31) return hipErrorInvalidValue;
    const long long total = rows * (hidden / 4);
    hipLaunchKernelGGL(geglu_planes_kernel, dim3(blocks_for(total, 4096)), dim3(256), 0, s, proj, reinterpret_cast<unsigned char*>(out3), rows, hidden / 4);
    return hipGetLastError();
}
hipError_t launch_geglu(const float* proj, float* out, long long rows, int hidden, hipStream_t s) {
    if (hidden & 3) return hipErrorInvalidValue;
    const long long total = rows * (hidden / 4);
    hipLaunchKernelGGL(geglu_kernel, dim3(blocks_for(total, 4096)), dim3(256), 0, s, proj, out, rows, hidden / 4);
    return hipGetLastError();
}
// dst[i][:] = src[:] for i < n (elems % 4 == 0): the prompt embedding replicated over the images of a shard (sample/main.rs:100-109)
__global__ void repeat_rows_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int n, long long elems4) {
    const long long total = elems4 * n;
    GRID_STRIDE(i, total) dst[i] = src[i % elems4];
}
hipError_t launch_repeat_rows(const float* src, float* dst, int n, long long elems, hipStream_t s) {
    if (elems & 3) return hipErrorInvalidValue;
    if (n <= 0 || elems <= 0) return hipSuccess;
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(blocks_for(elems / 4 * n)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), n, elems / 4);
    return hipGetLastError();
}
hipError_t launch_silu(const float* x, float* y, long long n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}
hipError_t launch_nchw3_to_nhwc4(const float* src, float* dst, int n, int h, int w, hipStream_t s) {
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks_for((long long)n * hw)), dim3(256), 0, s, src, dst, n, hw);
    return hipGetLastError();
}
hipError_t launch_quick_gelu(float* x, long long n, hipStream_t s) {
    hipLaunchKernelGGL(quick_gelu_kernel, dim3(blocks_for(n)), dim3(256), 0, s, x, n);
    return hipGetLastError();
}
hipError_t launch_clip_embed(const int* tokens, const float* tok_table, const float* pos_table, float* out, int n, int T, int C,
                             hipStream_t s) {
    if (C % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clip_embed_kernel, dim3(blocks_for((long long)n * T * (C / 4))), dim3(256), 0, s, tokens, tok_table, pos_table,
                       out, n, T, C);
    return hipGetLastError();
}
hipError_t launch_causal_mask(float* mask, int T, hipStream_t s) {
    hipLaunchKernelGGL(causal_mask_kernel, dim3(blocks_for((long long)T * T)), dim3(256), 0, s, mask, T);
    return hipGetLastError();
}
hipError_t launch_transpose2d(const float* src, float* dst, int rows, int cols, int src_ld, hipStream_t s) {
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, s, src, dst, rows, cols, src_ld);
    return hipGetLastError();
}
hipError_t launch_timestep_embedding(const int* t_dev, int n_t, int dim, float* out, hipStream_t s) {
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(blocks_for((long long)n_t * dim / 2)), dim3(256), 0, s, t_dev,
                       n_t, dim, out);
    return hipGetLastError();
}
hipError_t launch_cfg_ddim(const float* eps, float* latent, float* unet_in, long long per_half, DdimCoef c,
                           hipStream_t s) {
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(blocks_for(per_half)), dim3(256), 0, s, eps, latent, unet_in, per_half,
                       c);
    return hipGetLastError();
}
hipError_t launch_dup_latent(const float* latent, float* unet_in, long long per_half, hipStream_t s) {
    hipLaunchKernelGGL(dup_latent_kernel, dim3(blocks_for(per_half)), dim3(256), 0, s, latent, unet_in, per_half);
    return hipGetLastError();
}
hipError_t launch_image_to_u8(const float* img_nhwc, uint8_t* out, long long n_elem, hipStream_t s) {
    hipLaunchKernelGGL(image_to_u8_kernel, dim3(blocks_for(n_elem, 4096)), dim3(256), 0, s, img_nhwc, out, n_elem);
    return hipGetLastError();
}
hipError_t launch_fill_normal(float* dst, long long n, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_normal_kernel, dim3(blocks_for(n)), dim3(256), 0, s, dst, n, seed);
    return hipGetLastError();
}

}  // namespace sdmi
