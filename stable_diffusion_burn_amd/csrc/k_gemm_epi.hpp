// k_gemm_epi.hpp -- the fp32 output epilogue shared by the 8-wave large-tile kernels whose accumulators have the
// 16x16 MFMA D layout (lane (c = lane & 15, g = lane >> 4) holds D[4g .. 4g+3][c], D rows = output channels, D columns =
// pixels): k_gemm2x.hip (v_mfma_f32_16x16x4_f32) and k_gemm3x.hip (v_mfma_f32_16x16x32_bf16 on split operands).
//   * bias + time-embedding row + residual, or the GEGLU product, or a raw split-K slab
//   * every wave transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free by then)
//     and writes whole row segments with 16-byte lanes; the residual is read the same way
//   * split-K: raw fp32 slabs, combined by launch_splitk_reduce in a second launch -- or, with p.csk, inside this launch by the workgroups
//     that produced them (csk_combine below)
#pragma once
#include "kernels.hpp"
#include "k_common.hpp"
#include "k_split3.hpp"

namespace sdmi {

typedef float epi_f32x4 __attribute__((ext_vector_type(4)));

// ---- split-K combined inside the launch, cooperatively -------------------------------------------------------------------------------------
// The S slices of an output tile are S workgroups with the same blockIdx.x, i.e. (gemm_grid: gridDim.x is a multiple of 8, workgroups are dealt
// round-robin to the 8 XCDs in linear order, starting at XCD 0) S workgroups of ONE XCD: their slab tiles meet in that XCD's L2 and need no write-back to be visible
// to each other.  Each slice stores its partial tile, announces itself on the tile's arrival word and, once all S have arrived, sums rows
// [z R, (z + 1) R) of the tile over the slabs IN SLICE ORDER (bit-reproducible whatever the arrival order), applies bias / time-embedding row /
// residual and writes C and / or the planes C3: every slice reduces 1 / S of the tile, nothing is read by a workgroup that did not just write
// 1 / S of it, and the separate reduce launch with its two kernel boundaries is gone.
//   arrival word (64 bit, zero at launch): bits 0-7 arrivals, 32-63 slices that gave up waiting (their rows are summed by the slice that arrives last).
//   * nobody depends on co-residency: a slice polls the word a bounded number of times, then sets its bit and leaves; the two atomics (arrive:
//     add, give up: or) are totally ordered on the word, so either the last arriver sees the bit in the value its add returns and does those
//     rows too, or the or returns a full count and the slice does its rows itself.
//   * the same-XCD premise is CHECKED, not assumed: every workgroup compares its XCC id (s_getreg HW_REG_XCC_ID) with blockIdx.x % 8 and sets
//     *csk_flag otherwise; the engine then fails the call (Engine::check_csk_flag) -- stale slab bytes are never returned silently -- and a
//     self-test at context creation (launch_xcc_selftest) switches the option off where the dispatcher deals differently (partitioned modes).
//   * the words are zero again for the next launch without a memset: two arrays alternate, every launch zeroes the other one (the previous
//     launch that used it has completed: same stream).
__device__ __forceinline__ unsigned csk_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 15u;
}

// rows [mr0, mr0 + rows) x 16-byte columns [n0, n0 + 4 cpr) of the output: sum of the S slabs in slice order + bias / row vector / residual -> C, C3.
// U outputs per thread per pass, J slabs in flight per output.
template <int U, int J>
__device__ __forceinline__ void csk_sum(const ConvGemm& p, const int mr0, const int n0, const int rows, const int cpr, const int HoWo) {
    typedef epi_f32x4 f32x4;
    const int S = p.splits;
    const int total = rows * cpr;
    for (int e0 = threadIdx.x; e0 < total; e0 += U * (int)blockDim.x) {
        long long off[U];
        int mm[U], nn[U];
        bool ok[U];
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * (int)blockDim.x;
            ok[u] = e < total;
            const int ee = ok[u] ? e : 0;
            const int r = ee / cpr;
            mm[u] = mr0 + r;
            nn[u] = n0 + ((ee - r * cpr) << 2);
            off[u] = (long long)mm[u] * p.N + nn[u];
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int s = 0; s < S; s += J) {
            f32x4 t[J][U];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int sj = min(s + j, S - 1);         // past the end: a harmless re-read, not added
#pragma unroll
                for (int u = 0; u < U; ++u) t[j][u] = *reinterpret_cast<const f32x4*>(p.slabs + (long long)sj * p.slab_stride + off[u]);
            }
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (s + j < S) {
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = (s + j == 0) ? t[j][u] : v[u] + t[j][u];
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            f32x4 r = v[u];
            const int m = mm[u], n = nn[u];
            if (p.bias) r += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.rowvec) r += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)(m / HoWo) * p.rowvec_stride + n);
            if (p.resid) r += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
            if (p.C) *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = r;
            if (p.C3) s3_store4(reinterpret_cast<unsigned char*>(p.C3) + (long long)m * p.ldc3, n, r);
        }
    }
}

template <int BM, int BN>
__device__ __forceinline__ void csk_combine(const ConvGemm& p, unsigned char* smem, const int m0, const int n0, const int z, const int lid,
                                            const int MT_NT, const int HoWo) {
    typedef epi_f32x4 f32x4;
    const int tid = threadIdx.x;
    const int S = p.splits;
    // zero the other array for the launch after this one (slice 0 of every tile: words lid, lid + tiles, ...)
    if (z == 0)
        for (int j = lid + tid * MT_NT; j < kCskWords; j += (int)blockDim.x * MT_NT) p.csk_other[j] = 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's slab stores are in the L2
    __syncthreads();
    unsigned long long* const pd = p.probe ? p.probe + 24ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.z) : nullptr;   // (diagnostic builds)
    if (pd && tid == 0) pd[20] = __builtin_amdgcn_s_memrealtime();
    unsigned* flags = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) {
        unsigned long long* W = p.csk + lid;
        if (csk_xcc_id() != (blockIdx.x & 7u)) __hip_atomic_store(p.csk_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long old = __hip_atomic_fetch_add(W, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned mine = 1u, others = 0u;
        if ((int)(old & 0xFFull) + 1 == S) {
            others = (unsigned)(old >> 32);               // I am last: the rows of every slice that gave up before I arrived are mine too
        } else {
            unsigned long long w;
            int it = 0;
            do {
                __builtin_amdgcn_s_sleep(4);
                w = __hip_atomic_load(W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((int)(w & 0xFFull) != S && ++it < 1000);
            if ((int)(w & 0xFFull) != S) {
                const unsigned long long o2 = __hip_atomic_fetch_or(W, 1ull << (32 + z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(o2 & 0xFFull) != S) mine = 0u;  // the slice that arrives last will see my bit
            }
        }
        flags[0] = mine;
        flags[1] = others;
    }
    __syncthreads();
    const unsigned mine = flags[0], others = flags[1];
    if (pd && tid == 0) pd[21] = __builtin_amdgcn_s_memrealtime();
    if (!(mine | others)) return;
    // No cache maintenance here, on purpose.  The other slices' slab tiles are read through this CU's L1, which cannot hold them: it was
    // invalidated when the kernel was dispatched, this workgroup has not read slab memory before, a tile's slab region consists of whole cache
    // lines (launch side: N % 32 == 0) and nobody reads a tile's region before all of its slices have arrived -- so the reads miss to the XCD's
    // L2, where the writers' stores are (they waited for vmcnt(0) before arriving).  An agent-scope acquire (buffer_inv sc1) in this place
    // measured 8 us per workgroup and slowed the loads behind it by another 5 (profiles/r03p_*): more than the launch it replaces.
    if (pd) { __syncthreads(); if (tid == 0) pd[22] = __builtin_amdgcn_s_memrealtime(); }
    const int R = (BM + S - 1) / S;
    const int cols = min(BN, p.N - n0);
    const int cpr = cols >> 2;                            // 16-byte outputs per row (launch side: N % 32 == 0)
    for (int q = 0; q < S; ++q) {
        if (!((q == z && mine) || ((others >> q) & 1u))) continue;
        const int r0 = q * R;
        const int rows = min(min(R, BM - r0), p.M - m0 - r0);
        if (rows <= 0 || cpr <= 0) continue;
        // a thread's (outputs) x (slabs) loads are all in flight together: ~20 per thread whatever S is (rows shrink as S grows)
        const int per_thread = (rows * cpr + (int)blockDim.x - 1) / (int)blockDim.x;
        if (per_thread >= 4) csk_sum<5, 4>(p, m0 + r0, n0, rows, cpr, HoWo);
        else if (per_thread == 3) csk_sum<3, 8>(p, m0 + r0, n0, rows, cpr, HoWo);
        else if (per_thread == 2) csk_sum<2, 12>(p, m0 + r0, n0, rows, cpr, HoWo);
        else csk_sum<1, 16>(p, m0 + r0, n0, rows, cpr, HoWo);
    }
    if (pd) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (tid == 0) pd[23] = __builtin_amdgcn_s_memrealtime(); }
}

template <int MI, int NI, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_f32(const ConvGemm& p, epi_f32x4 (&acc)[MI][NI], unsigned char* smem_x32, const int m0,
                                                  const int n0, const int z, const int lid, const int wave, const int lane,
                                                  const int HoWo) {
    typedef epi_f32x4 f32x4;
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    const bool geglu = p.geglu != 0;
    // ---- epilogue: bias + time-embedding row + residual, fp32 -------------------------------------------
    // Each wave transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free
    // now) and writes whole 320-byte row segments with 16-byte lanes; the residual is read the same way.
    const bool split = p.splits > 1;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = !split && p.resid;
    const bool vec_ok = ((p.N & 3) == 0) && ((ldc & 3) == 0) && ((p.ldr & 3) == 0 || !has_resid);
    constexpr int LDSW = WNC + 4;       // scratch row stride in floats
    if (geglu) {   // launch-side guarantees: NI even, no split-K, N % 8 == 0, ldc % 8 == 0, no rowvec / residual
        if constexpr (NI % 2 == 0) {
            constexpr int WNO = WNC / 2;     // output columns of a wave tile
            constexpr int LDSW2 = WNO + 4;
            __syncthreads();
            float* scr = reinterpret_cast<float*>(smem_x32 + wave * (16 * LDSW2 * 4));
            const int nw0 = n0 + wn * WNO;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
#pragma unroll
                for (int j = 0; j < NI / 2; ++j) {
                    const int n = nw0 + j * 16 + g4 * 4;
                    f32x4 v = acc[mi][2 * j], g = acc[mi][2 * j + 1];
                    if (p.bias && n < p.N) {
                        v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        g += *reinterpret_cast<const f32x4*>(p.bias + p.N + n);
                    }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[e] * (0.5f * g[e] * (1.0f + erff(g[e] * 0.70710678118654752440f)));
                    *reinterpret_cast<f32x4*>(scr + c15 * LDSW2 + j * 16 + g4 * 4) = o;
                }
                __builtin_amdgcn_wave_barrier();
                constexpr int CH = WNO / 4;
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c4 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c4 * 4;
                    if (q < 16 * CH && m < p.M && n < p.N)
                        *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c4 * 4);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    if (vec_ok) {
        __syncthreads();                // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_x32 + wave * (16 * LDSW * 4));
        const int nw0 = n0 + wn * WNC;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mrow0 = m0 + (wm * MI + mi) * 16;
            {
                const int m = mrow0 + c15;
                const int smp = (m < p.M ? m : 0) / HoWo;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = nw0 + ni * 16 + g4 * 4;
                    f32x4 v = acc[mi][ni];
                    if (!split && n < p.N) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    }
                    *reinterpret_cast<f32x4*>(scr + c15 * LDSW + ni * 16 + g4 * 4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            constexpr int CH = WNC / 4;   // 16-byte chunks per row
#pragma unroll
            for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                const int q = q0 + lane;
                const int row = q / CH, c4 = q - row * CH;
                const int m = mrow0 + row, n = nw0 + c4 * 4;
                if (q < 16 * CH && m < p.M && n < p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                    if (has_resid) v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                    if (split) *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;     // (Cf = this k slice's slab)
                    else {
                        if (Cf) *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;
                        if (p.C3) s3_store4(reinterpret_cast<unsigned char*>(p.C3) + (long long)m * p.ldc3, n, v);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
    // odd strides / N not a multiple of 4: element-wise stores straight from the accumulators
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        if (m >= p.M) continue;
        const int smp = m / HoWo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 16 + g4 * 4;
            if (n >= p.N) continue;
            const f32x4 v = acc[mi][ni];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < p.N) {
                    float sv = v[r];
                    if (!split) {
                        if (p.bias) sv += p.bias[n + r];
                        if (p.rowvec) sv += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                        if (p.resid) sv += p.resid[(long long)m * p.ldr + n + r];
                    }
                    Cf[(long long)m * ldc + n + r] = sv;
                }
            }
        }
    }
    }
    if (split && p.csk) csk_combine<BM, BN>(p, smem_x32, m0, n0, z, lid, ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), HoWo);
}

}  // namespace sdmi
