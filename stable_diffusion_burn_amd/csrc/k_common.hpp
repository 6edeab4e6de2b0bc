// k_common.hpp -- device helpers shared by several kernel translation units of libsdmi.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sdmi {

// Round 6.  A workgroup barrier that PUBLISHES LDS-DMA data (global_load_lds) is only correct if every wave waits for ITS OWN outstanding pieces before it enters the
// barrier -- vmcnt is per wave, and the pieces a wave reads behind the barrier were issued by other waves.  __syncthreads() alone does not promise that wait: on gfx950
// a workgroup-scope fence needs no vmcnt(0), and hipcc places the wait it derives for the LDS-DMA -> ds_read dependence wherever its pass decides -- in front of the
// barrier in every kernel of rounds 1-5, BEHIND it in the persistent GEGLU instantiation of k_gemm_bf16x.hip once its loop head changed (profiles/r06i: batch-32
// forwards differing by 3e-2 from run to run).  Every k-loop barrier of the LDS-DMA kernels therefore spells the wait out; tests/test_code_objects_cpu.py checks the ISA.
__device__ __forceinline__ void sdmi_dma_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// ---- which (M tile, N tile, split-K slice) a GEMM block works on ------------------------------------------------------
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; a speed assumption only, never a correctness one:
// every work item is covered exactly once whatever the placement): an XCD works on a contiguous band of tiles, blockIdx.z is the slice.
// (Rounds 2-3 also carried a planner that cut M tiles x N tiles x slices over the XCDs to minimise the bytes crossing the fabric: 3-6x
// fewer L2 misses on the batch-1 shapes and 2 % SLOWER end to end in both rounds -- profiles/r02zz_*, r03m_ab_fp32_b1_xcd_map.jsonl --
// because the duplicate fetches it saves are Infinity-Cache hits; removed.)
struct GemmWork { int tm, tn, z, lid; bool live; };
__device__ __forceinline__ GemmWork gemm_work_of_block(const ConvGemm& p, int MT, int NT) {
    GemmWork w;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tpx = gridDim.x >> 3;
    const int lid = x * tpx + j;
    w.tm = lid / NT;
    w.tn = lid - w.tm * NT;
    w.z = blockIdx.z;
    w.live = lid < MT * NT;
    w.lid = lid;
    return w;
}

// ---- the GEGLU gate's GELU (unet/mod.rs:587-590: x * 0.5 * (1 + erf(x / sqrt 2))) for the REDUCED-PRECISION kernels (bf16 / MXFP8 results) ---------------------------------
// erff() is ~ 35 instructions and two divergent branches per element; the fused bf16 GEGLU epilogue evaluates it 64 times per thread per 256 x 256 tile -- ~ 5 us beside a
// 6 us matrix loop (round 5).  Here: Phi(g) = 0.5 erfc(-g / sqrt 2) with Abramowitz-Stegun 7.1.26 (erfc(x) = t (a1 + t (a2 + ... a5 t)) exp(-x^2), t = 1 / (1 + p x),
// |error| <= 1.5e-7) on the hardware reciprocal and exp2: 17 instructions, no branch.  |gate - exact| <= 4.7e-7 over |g| <= 12 (tests/test_split_oracle_cpu.py), i.e.
// far inside the bf16 rounding of the result (1.4 % of the results land on the neighbouring bf16 value, always the adjacent one).  The fp32 kernels keep erff().
__device__ __forceinline__ float gelu_gate_fast(float g) {
    const float x = __builtin_fabsf(g) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float pl = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    pl = __builtin_fmaf(pl, t, 0.5f * 1.421413741f);
    pl = __builtin_fmaf(pl, t, 0.5f * -0.284496736f);
    pl = __builtin_fmaf(pl, t, 0.5f * 0.254829592f);
    const float h = (pl * t) * __builtin_amdgcn_exp2f((g * g) * -0.72134752044448170368f);    // 0.5 erfc(|g| / sqrt 2)
    return g * (0.5f + __builtin_copysignf(0.5f - h, g));
}

// Two gates at once (round 6): the same operations on <2 x float>, which gfx950 issues as v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 -- one issue slot per PAIR for 13 of
// the 17 operations (the reciprocals and exponentials stay scalar).  Each half is computed exactly as gelu_gate_fast computes it: bit-identical results.
typedef float kc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ kc_f32x2 gelu_gate_fast2(const kc_f32x2 g) {
    const kc_f32x2 ag = kc_f32x2{__builtin_fabsf(g[0]), __builtin_fabsf(g[1])};
    const kc_f32x2 x = ag * 0.70710678118654752440f;
    const kc_f32x2 den = __builtin_elementwise_fma(kc_f32x2{0.3275911f, 0.3275911f}, x, kc_f32x2{1.0f, 1.0f});
    const kc_f32x2 t = kc_f32x2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    kc_f32x2 pl = __builtin_elementwise_fma(t, kc_f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, kc_f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
    pl = __builtin_elementwise_fma(pl, t, kc_f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
    pl = __builtin_elementwise_fma(pl, t, kc_f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
    pl = __builtin_elementwise_fma(pl, t, kc_f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
    const kc_f32x2 ex = (g * g) * -0.72134752044448170368f;
    const kc_f32x2 h = (pl * t) * kc_f32x2{__builtin_amdgcn_exp2f(ex[0]), __builtin_amdgcn_exp2f(ex[1])};    // 0.5 erfc(|g| / sqrt 2)
    const kc_f32x2 d = 0.5f - h;
    const kc_f32x2 sd = kc_f32x2{__builtin_copysignf(d[0], g[0]), __builtin_copysignf(d[1], g[1])};
    return g * (0.5f + sd);
}

// ---- GroupNorm statistics (groupnorm/mod.rs:75-82) ---------------------------------------------------------
// Partial statistics of one (sample, chunk, group) are (mean, M2 = sum (x - mean)^2) over the chunk's rows x (C/G)
// channels: part[((smp*chunks + chunk)*G + g)*2 + {0, 1}].  Producers: gn_stats_kernel / gn_stats_bf16_kernel and the
// GEMM epilogues that emit them for their own output tile.
// Tail of the statistics kernels: per-channel (mean_c, M2_c) of one chunk (equal counts n_rows) -> per-group (mean, M2), shifted
// by the group's first channel mean.  Four threads per group walk every fourth channel and are combined with two xor-shuffles
// in a fixed order (a 3-sum fp64 dependency chain over up to 80 channels in ONE thread cost 1 us per launch).
__device__ __forceinline__ double kc_shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void gn_merge_group_channels(const double* chm, const double* chq, int C, int G, double n_rows, double* out) {
    const int cpg = C / G;
    const double inv_cpg = 1.0 / cpg;
    if ((blockDim.x & 3) == 0) {
        for (int idx = threadIdx.x; idx < G * 4; idx += blockDim.x) {
            const int gi = idx >> 2, j = idx & 3;
            const double ref = chm[gi * cpg];
            double a = 0.0, b = 0.0, m2 = 0.0;
            for (int ch = gi * cpg + j; ch < (gi + 1) * cpg; ch += 4) {
                const double d = chm[ch] - ref;
                a += d; b += d * d; m2 += chq[ch];
            }
            a += kc_shfl_xor_f64(a, 1); b += kc_shfl_xor_f64(b, 1); m2 += kc_shfl_xor_f64(m2, 1);
            a += kc_shfl_xor_f64(a, 2); b += kc_shfl_xor_f64(b, 2); m2 += kc_shfl_xor_f64(m2, 2);
            if (j == 0) {
                out[gi * 2] = ref + a * inv_cpg;
                out[gi * 2 + 1] = m2 + n_rows * (b - a * a * inv_cpg);
            }
        }
    } else {
        for (int gi = threadIdx.x; gi < G; gi += blockDim.x) {
            const double ref = chm[gi * cpg];
            double a = 0.0, b = 0.0, m2 = 0.0;
            for (int ch = gi * cpg; ch < (gi + 1) * cpg; ++ch) {
                const double d = chm[ch] - ref;
                a += d; b += d * d; m2 += chq[ch];
            }
            out[gi * 2] = ref + a * inv_cpg;
            out[gi * 2 + 1] = m2 + n_rows * (b - a * a * inv_cpg);
        }
    }
}

// Merges the chunk partials of every group (8 threads per group, fixed order, one shifted pass in fp64) into
// mean (as a float-float pair) and 1/sqrt(var + eps).  stat_rows = rows per partial chunk (the last may be short).
__device__ __forceinline__ void gn_finalize(const double* __restrict__ part, int smp, int G, int cpg, int hw, int stat_chunks,
                                            int stat_rows, float eps, double (*s_red)[64][8], float* s_mean_hi, float* s_mean_lo,
                                            float* s_rstd) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < G * 8; idx += blockDim.x) {
        const int gi = idx >> 3, j = idx & 7;
        const double* pp = part + ((long long)smp * stat_chunks * G + gi) * 2;
        const double ref = pp[0];
        double a = 0.0, b = 0.0, m2 = 0.0;
        for (int ch = j; ch < stat_chunks; ch += 8) {
            const double cnt = (double)(min((ch + 1) * stat_rows, hw) - ch * stat_rows) * cpg;
            const double d = pp[(long long)ch * G * 2] - ref;
            a += cnt * d; b += cnt * d * d; m2 += pp[(long long)ch * G * 2 + 1];
        }
        s_red[0][gi][j] = a;
        s_red[1][gi][j] = b;
        s_red[2][gi][j] = m2;
    }
    __syncthreads();
    for (int gi = tid; gi < G; gi += blockDim.x) {
        double a = 0.0, b = 0.0, m2 = 0.0;
        for (int j = 0; j < 8; ++j) { a += s_red[0][gi][j]; b += s_red[1][gi][j]; m2 += s_red[2][gi][j]; }
        const double inv_cnt = 1.0 / ((double)hw * cpg);
        const double mean = part[((long long)smp * stat_chunks * G + gi) * 2] + a * inv_cnt;
        double var = (m2 + b - a * a * inv_cnt) * inv_cnt;
        if (var < 0.0) var = 0.0;
        const float hi = (float)mean;
        s_mean_hi[gi] = hi;
        s_mean_lo[gi] = (float)(mean - (double)hi);
        s_rstd[gi] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
}


}  // namespace sdmi
