// k_gemm.hip -- what the implicit-GEMM convolution / linear kernels share: the math and mapping notes below, the tile
// table, the weight packing kernels and the stand-alone split-K reduce kernel.  The GEMM kernels themselves are
// k_gemm2.hip (4 waves, register staged), k_gemm2x.hip (8 waves, LDS-DMA) and their bf16 twins.
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
//  * 256-thread workgroup = 4 waves (one per SIMD), wave tile (16*MI) x (16*NI); LDS tiles and pipeline: k_gemm2.hip.
//  * k order = (channel slice, tap, 32 channels): the 9 taps of a slice re-read
//    the same 128-byte pixel segments, which stay in L1/L2.
//  * blockIdx -> tile map is XCD aware: blocks that land on one XCD (bid % 8)
//    own a contiguous range of tiles (n fastest), so an XCD's L2 keeps its band
//    of the activation and its neighbours' halo rows.
//  * split-K over blockIdx.z writes raw fp32 slabs; launch_splitk_reduce sums them in fixed order (bit-reproducible)
//    and applies the epilogue.  Three other places for the combine were built and measured slower (profiles/README.md): inside the launch by
//    the last-arriving slice (round 2), inside the GroupNorm / LayerNorm that reads the result, and inside the launch by all slices of one XCD
//    (round 3) -- the combine moves (slices + 1) x the tensor between the L2s and the Infinity Cache in every form, and this kernel does it
//    at the rate that traffic allows without making any GEMM workgroup wait.
#include "kernels.hpp"
#include "k_split3.hpp"

#include <mutex>
#include <set>
#include <utility>

namespace sdmi {

// Kernels that need more than 64 KiB of dynamic LDS must say so once per (kernel, device).  The engines of a multi-device context
// launch from one host thread per device (multi.cpp), so the "already done" set is keyed by the current device and guarded.
hipError_t set_max_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- split-K reduction + epilogue ------------------------------------------------
// G lanes share one 16-byte output: lane g sums slabs g, g + G, g + 2G, ... in that order, then the G partial sums are
// combined by xor-shuffles (1, 2, 4) -- a fixed order, so results stay bit-reproducible.  G > 1 is for small outputs with
// many slabs (M = 128 ... 512 rows x 16 ... 32 slices at batch 1): one thread per output would leave most of the chip idle
// behind a serial chain of 32 dependent-latency loads.
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvGemm p) {
    const float* slabs = p.slabs;
    float* C = p.C;
    const int HoWo = p.Ho * p.Wo;
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!p.resid || (p.ldr & 3) == 0);
    if (vec_ok) {
        const int n4 = p.N >> 2;
        const long long total = (long long)p.M * n4;              // 16-byte outputs
        const long long rounded = (total * G + 255) / 256 * 256;   // whole workgroups take part in the shuffles
        for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < rounded; t += (long long)gridDim.x * blockDim.x) {
            const long long i = t / G;
            const int g = (int)(t - i * G);
            const bool live = i < total;
            const long long ii = live ? i : 0;
            const int m = (int)(ii / n4);
            const int n = (int)(ii - (long long)m * n4) * 4;
            const long long off = (long long)m * p.N + n;
            // round 6: the bias / time-embedding row / residual of this output are requested BEFORE the slab loads, not behind the shuffles -- the launch is one memory round
            // trip deep instead of two (2 427 launches of 6 ... 10 us per batch-1 image); added in the same order as before: bit-identical results
            f32x4 eb = {0.f, 0.f, 0.f, 0.f}, ev = eb, er = eb;
            if (live && g == 0) {
                if (p.bias) eb = *reinterpret_cast<const f32x4*>(p.bias + n);
                if (p.rowvec) ev = *reinterpret_cast<const f32x4*>(p.rowvec + (long long)(m / HoWo) * p.rowvec_stride + n);
                if (p.resid) er = *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            int s = g;
            for (; s + 3 * G < p.splits; s += 4 * G) {      // four loads in flight, summed in slice order
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (long long)s * p.slab_stride + off);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(slabs + (long long)(s + G) * p.slab_stride + off);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 2 * G) * p.slab_stride + off);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(slabs + (long long)(s + 3 * G) * p.slab_stride + off);
                v += a0; v += a1; v += a2; v += a3;
            }
            for (; s < p.splits; s += G) v += *reinterpret_cast<const f32x4*>(slabs + (long long)s * p.slab_stride + off);
#pragma unroll
            for (int o = 1; o < G; o <<= 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
            }
            if (live && g == 0) {
                if (p.bias) v += eb;
                if (p.rowvec) v += ev;
                if (p.resid) v += er;
                if (C) *reinterpret_cast<f32x4*>(C + (long long)m * p.ldc + n) = v;
                if (p.C3) s3_store4(reinterpret_cast<unsigned char*>(p.C3) + (long long)m * p.ldc3, n, v);
            }
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

hipError_t launch_splitk_reduce(const ConvGemm& p, hipStream_t stream) {
    const bool vec = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!p.resid || (p.ldr & 3) == 0);
    if (!vec && (p.C3 || !p.C)) return hipErrorInvalidValue;   // the plane output is part of the 16-byte path only
    const long long work = ((long long)p.M * p.N + 3) / 4;
    // lanes per output: enough threads to cover the chip (>= 256 K) while every lane still has two slabs to sum
    int g = 1;
    if (vec)
        while (g < 8 && work * g < 262144 && 4 * g <= p.splits) g *= 2;
    long long blocks = (work * g + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    switch (g) {
        case 1: hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, stream, p); break;
    }
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
