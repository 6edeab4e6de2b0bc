// exp_mfma_probe.hip -- what v_exp_f32 costs on gfx950 and whether it overlaps the matrix pipe (round 4; the bf16 attention at d = 40 spends as many
// cycles as its matrix and transcendental work ADDED).  Shader-clock cycles per loop iteration of one wave, for:
//   mode 0: 64 v_exp_f32                       mode 1: 28 v_mfma_f32_32x32x16_bf16 (4 accumulators)      mode 2: both, interleaved in ONE wave (7 groups of 4 MFMA + ~9 exp)
//   mode 3: 64 v_cvt_pk_bf16_f32               mode 4: 64 v_max3_f32                                       mode 5: 64 v_exp_f16 ... (see below)
// with 1 wave per SIMD (256 threads) and 2 waves per SIMD (512 threads); and mode 6: 512 threads, waves 0-3 run mode 0 and waves 4-7 mode 1 (one of each per SIMD).
// build: hipcc -O3 --offload-arch=gfx950 exp_mfma_probe.hip -o exp_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define EXP4(a, b, c, d) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define CVT4(a, b, c, d) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %1, %2\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define MAX4(a, b, c, d) asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %0\n\tv_max3_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define FMA4(a, b, c, d) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

__global__ __launch_bounds__(512) void probe(int mode, int iters, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = -1.0f - 0.01f * (threadIdx.x + i);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.5f); }
    int m = mode;
    if (mode == 6) m = wave < 4 ? 0 : 1;
    if (mode == 7) m = wave < 4 ? 4 : 1;     // plain VALU beside the matrix wave
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (m == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { EXP4(x[0], x[1], x[2], x[3]); EXP4(x[4], x[5], x[6], x[7]); EXP4(x[8], x[9], x[10], x[11]); EXP4(x[12], x[13], x[14], x[15]); }
        } else if (m == 1) {
#pragma unroll
            for (int g = 0; g < 7; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        } else if (m == 2) {
#pragma unroll
            for (int g = 0; g < 7; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(x[(4 * g + j) & 15]), "+v"(x[(4 * g + j + 7) & 15]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            EXP4(x[0], x[1], x[2], x[3]); EXP4(x[4], x[5], x[6], x[7]);   // 56 + 8 = 64
        } else if (m == 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { CVT4(x[0], x[1], x[2], x[3]); CVT4(x[4], x[5], x[6], x[7]); CVT4(x[8], x[9], x[10], x[11]); CVT4(x[12], x[13], x[14], x[15]); }
        } else if (m == 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { MAX4(x[0], x[1], x[2], x[3]); MAX4(x[4], x[5], x[6], x[7]); MAX4(x[8], x[9], x[10], x[11]); MAX4(x[12], x[13], x[14], x[15]); }
        } else if (m == 5) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { FMA4(x[0], x[1], x[2], x[3]); FMA4(x[4], x[5], x[6], x[7]); FMA4(x[8], x[9], x[10], x[11]); FMA4(x[12], x[13], x[14], x[15]); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += x[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[j][0] + acc[j][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 2000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * 8 * sizeof(long long));
    const char* names[] = {"64 v_exp_f32", "28 mfma 32x32x16 bf16", "28 mfma + 64 v_exp interleaved, one wave", "64 v_cvt_pk_bf16_f32", "64 v_max3_f32", "64 v_fma_f32",
                           "waves 0-3: 64 v_exp | waves 4-7: 28 mfma", "waves 0-3: 64 v_max3 | waves 4-7: 28 mfma"};
    for (int mode = 0; mode < 8; ++mode)
        for (int threads : {256, 512}) {
            if (mode >= 6 && threads == 256) continue;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, 10, out, cyc);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, mode, iters, out, cyc);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * 8);
            hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            const int nw = threads / 64;
            double lo = 0, hi = 0;   // mean over blocks of waves 0..3 and of waves 4..7 (or the same set at 4 waves)
            for (int bI = 0; bI < blocks; ++bI)
                for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += (double)h[bI * nw + w];
            lo /= blocks * 4.0 * iters;
            hi = nw > 4 ? hi / (blocks * 4.0 * iters) : 0.0;
            printf("mode %d  %-48s %d waves/SIMD: %8.1f cycles/iteration (waves 0-3)", mode, names[mode], nw / 4, lo);
            if (nw > 4) printf("  %8.1f (waves 4-7)", hi);
            printf("   %.3f ms -> %.2f GHz-equivalent\n", ms, lo * iters / (ms * 1e6));
        }
    return 0;
}
