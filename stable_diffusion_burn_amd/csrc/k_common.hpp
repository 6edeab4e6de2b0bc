// k_common.hpp -- device helpers shared by several kernel translation units of libsdmi.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sdmi {

// ---- which (M tile, N tile, split-K slice) a GEMM block works on ------------------------------------------------------
// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed; a speed assumption only, never a correctness one:
// every work item is covered exactly once whatever the placement).  lid = tm * NT + tn indexes the tile's split-K arrival counter.
struct GemmWork { int tm, tn, z, lid; bool live; };
__device__ __forceinline__ GemmWork gemm_work_of_block(const ConvGemm& p, int MT, int NT) {
    GemmWork w;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (p.xcd_m > 0) {
        const int im = x % p.xcd_m;
        const int t = x / p.xcd_m;
        const int in = t % p.xcd_n;
        const int iz = t / p.xcd_n;
        const int mn = p.xcd_ml * p.xcd_nl;
        const int zl = j / mn;
        const int r = j - zl * mn;
        const int tml = r / p.xcd_nl;
        w.tm = im * p.xcd_ml + tml;
        w.tn = in * p.xcd_nl + (r - tml * p.xcd_nl);
        w.z = iz * p.xcd_zl + zl;
        w.live = (w.tm < MT) & (w.tn < NT) & (w.z < p.splits) & (zl < p.xcd_zl);
    } else {
        const int tpx = gridDim.x >> 3;
        const int lid = x * tpx + j;
        w.tm = lid / NT;
        w.tn = lid - w.tm * NT;
        w.z = blockIdx.z;
        w.live = lid < MT * NT;
    }
    w.lid = w.tm * NT + w.tn;
    return w;
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


// ---- split-K combined inside the launch ---------------------------------------------------------------------
// Every k slice of an output tile stores its fp32 partial tile into its slab (plain stores), then calls
// splitk_arrive(); the workgroup that arrives last at the tile's counter sums the slabs in slice order 0..S-1 (so
// the result does not depend on which slice finished last: bit-reproducible) and applies the epilogue.  The
// hand-off is the agent-scope release / acquire recipe of the CDNA guide (section 5, "In-launch split-K reduction"):
// per-wave vmcnt(0) -> barrier -> one lane: release fence + vmcnt(0) + relaxed agent fetch_add; last arriver: one
// acquire fence, barrier, plain loads.  No spin anywhere: nobody waits for another workgroup.  Counters are zero
// between launches: the last arriver resets its counter (the engine zeroes the array once at creation).
// write_through: the slab tile was stored with sc1 (write-through) stores, which are complete once vmcnt drains -- no
// release fence, i.e. no write-back sweep of the XCD's whole L2 (buffer_wbl2) in every workgroup's tail.
__device__ __forceinline__ bool splitk_arrive(unsigned* counters, int tile, int splits, unsigned* lds_flag, bool write_through) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!write_through) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait behind buffer_wbl2 (guide, G16 pitfall 12)
        }
        const unsigned old = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)(splits - 1);
        if (last) {
            __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *lds_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

typedef float kc_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int kc_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int kc_u32x4 __attribute__((ext_vector_type(4)));

// 16-byte store of a slab element: plain, or write-through (sc1: aux bit 4 of the raw buffer store) through a descriptor
// over this k slice's slab.  byte_off < 4 GiB (slabs are M*N*4 bytes; the launch side checks).
struct SlabStore {
    __amdgpu_buffer_rsrc_t rsrc;
    float* base;
    bool wt;
    __device__ __forceinline__ SlabStore(float* slab, long long elems, bool write_through)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(unsigned)(elems * 4), 0x00020000)), base(slab), wt(write_through) {}
    __device__ __forceinline__ void store(long long elem_off, kc_f32x4 v) const {
        if (wt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(kc_u32x4, v), rsrc, (int)(unsigned)(elem_off * 4), 0, 16);
        else *reinterpret_cast<kc_f32x4*>(base + elem_off) = v;
    }
};

__device__ __forceinline__ unsigned kc_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// The last arriver's job: C[m0.., n0..] = epilogue(sum over slices of slabs[s][m][n]) for one bm x bn tile.  Needs
// N % 4 == 0, ldc % 4 == 0, ldr % 4 == 0 (the launch side falls back to the separate reduce kernel otherwise).
// BF16: residual and (unless out_mode == 1) output are bf16.  16 independent 16-byte loads per thread are in
// flight per pass: a dependent chain of cross-XCD loads would cost ~1-2 us per link.
template <bool BF16>
__device__ __forceinline__ void splitk_reduce_tile(const ConvGemm& p, int m0, int n0, int bm, int bn) {
    const int rows = min(bm, p.M - m0), cols = min(bn, p.N - n0);
    if (rows <= 0 || cols <= 0) return;
    const int cpr = cols >> 2;
    const int total = rows * cpr;
    const int HoWo = p.Ho * p.Wo;
    const bool out_f32 = !BF16 || p.out_mode == 1;
    constexpr int U = 4;
    for (int q0 = threadIdx.x; q0 < total; q0 += U * blockDim.x) {
        long long off[U];
        int mm[U], nn[U];
        bool ok[U];
        kc_f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * blockDim.x;
            ok[u] = q < total;
            const int qq = ok[u] ? q : 0;
            const int r = qq / cpr;
            mm[u] = m0 + r;
            nn[u] = n0 + ((qq - r * cpr) << 2);
            off[u] = (long long)mm[u] * p.N + nn[u];
            v[u] = *reinterpret_cast<const kc_f32x4*>(p.slabs + off[u]);
        }
        for (int s = 1; s < p.splits; s += 4) {
            kc_f32x4 t[4][U];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = min(s + j, p.splits - 1);   // past the end: a harmless re-read, not added
#pragma unroll
                for (int u = 0; u < U; ++u) t[j][u] = *reinterpret_cast<const kc_f32x4*>(p.slabs + (long long)sj * p.slab_stride + off[u]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (s + j < p.splits) {
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] += t[j][u];
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            kc_f32x4 r = v[u];
            const int m = mm[u], n = nn[u];
            if (p.bias) r += *reinterpret_cast<const kc_f32x4*>(p.bias + n);
            if (p.rowvec) r += *reinterpret_cast<const kc_f32x4*>(p.rowvec + (long long)(m / HoWo) * p.rowvec_stride + n);
            if (p.resid) {
                if (BF16) {
                    const kc_u32x2 h = *reinterpret_cast<const kc_u32x2*>(reinterpret_cast<const unsigned short*>(p.resid) + (long long)m * p.ldr + n);
                    r[0] += __uint_as_float(h[0] << 16); r[1] += __uint_as_float(h[0] & 0xFFFF0000u);
                    r[2] += __uint_as_float(h[1] << 16); r[3] += __uint_as_float(h[1] & 0xFFFF0000u);
                } else {
                    r += *reinterpret_cast<const kc_f32x4*>(p.resid + (long long)m * p.ldr + n);
                }
            }
            if (out_f32) {
                *reinterpret_cast<kc_f32x4*>(p.C + (long long)m * p.ldc + n) = r;
            } else {
                kc_u32x2 o = {kc_bf16_bits(r[0]) | (kc_bf16_bits(r[1]) << 16), kc_bf16_bits(r[2]) | (kc_bf16_bits(r[3]) << 16)};
                *reinterpret_cast<kc_u32x2*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o;
            }
        }
    }
}

}  // namespace sdmi
