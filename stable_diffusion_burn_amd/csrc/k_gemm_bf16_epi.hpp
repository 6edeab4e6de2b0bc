// k_gemm_bf16_epi.hpp -- the epilogue shared by the large-tile bf16 GEMM kernels (k_gemm_bf16x.hip, k_gemm_bf16p.hip): fp32 bias +
// time-embedding row + (bf16) residual, then a bf16 or fp32 store (split-K: the k slice's fp32 slab), or the GEGLU gate.
// acc[mi][ni] is the 16x16 fragment of rows (wm MI + mi) 16 .., columns (wn NI + ni) 16 ..; lane (c = lane & 15, g = lane >> 4) holds
// columns 4 g .. 4 g + 3 of row c.  Every wave of the workgroup calls it exactly once (one __syncthreads() inside); smem_x is the
// kernel's LDS, free for reuse once every wave has passed that barrier and no LDS-DMA is in flight.
#pragma once
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef float bepi_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bepi_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bepi_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xbf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float xbf16_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ unsigned xf32_to_bf16_bits(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
typedef float bepi_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bepi_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of bf16 (a low, b high), round to nearest even: ONE v_cvt_pk_bf16_f32 (the integer form above is 4 instructions per value; the epilogue of a
// 256 x 320 tile packs 81 920 values -- profiles/r04ad_*: 8 of a short-K launch's 27 us per tile were this epilogue's instructions, not its stores)
__device__ __forceinline__ unsigned xpack_bf16x2(float a, float b) {
    const bepi_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bepi_bf16x2));
}

// The accumulators' initial value (round 6): zero, or -- ConvGemm::resid_acc, set by the engine for launches without split-K -- what the epilogue used to ADD from
// global memory: bit 0 = the (bf16) residual tile, bit 1 = bias + time-embedding row.  C = (bias + rowvec + R) + A B, read in the accumulators' own layout (lane (c, g) of
// fragment (mi, ni): row c, columns 4 g .. 4 g + 3: one 8-byte load of four bf16 / one 16-byte load of four floats) in FRONT of the k loop, where the latency hides
// behind the first k tile's DMA, instead of in the epilogue, where every 16-row fragment group waited out its own loads (profiles/r04ae: N = 320, K = 320 at
// M = 131 072: 48.9 us without a residual, 80 us with one; profiles/r06g: +2.6 % per bf16 image).  The epilogue then issues no global load at all and takes its
// no-residual paths (2-byte scratch, persistent tile loop).  Needs N % 4 == 0, ldr % 4 == 0, rowvec_stride % 4 == 0 (engine).
// GEGLU_T: -1 = p.geglu at run time, 0 / 1 = known at compile time (fragment ni even = value columns, odd = their gate columns; bias rows N apart).
// The residual tile of an INTERIOR tile on top of the per-column terms colv, its loads issued in straight-line batches (RBM fragment rows x NI 8-byte loads in flight: what
// the caller's registers allow) and consumed afterwards.  The general form below -- `if (with_v) load; if (with_r) load;` per fragment -- is compiled to one block per
// load with an s_waitcnt vmcnt(0) at its end: 40 serialized round trips per 256 x 320 tile (profiles/r06zp_*: M = 131 072, N = K = 320 took 79 us with a residual,
// 41 us without; 13 us is what the residual's bytes cost).  Batching THAT form (predicates and pointer selects for 40 loads up front) spilled 30 - 180 registers in the
// 8-wave kernels, so it stays as it is for edge tiles and time-embedding rows, and the batched form has no predicate at all: one address per fragment row.
template <int MI, int NI, int WM, int WN, int RBM>
__device__ __forceinline__ void gemm_acc_resid_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], const bepi_f32x4 (&colv)[NI], const int ncol0, const int m0,
                                                    const int wave, const int lane) {
    constexpr int RB = MI > RBM ? RBM : MI;
    static_assert(RB > 0 && MI % RB == 0, "whole batches");
    const int wm = wave / WN;
    const int c15 = lane & 15;
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
#pragma unroll
    for (int mb = 0; mb < MI / RB; ++mb) {
        bepi_u32x2 raw[RB * NI];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const unsigned short* row = Rh + (long long)(m0 + (wm * MI + mb * RB + i) * 16 + c15) * p.ldr + ncol0;     // one address per fragment row, the fragments at + 32 B
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) raw[i * NI + ni] = *reinterpret_cast<const bepi_u32x2*>(row + ni * 16);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const bepi_u32x2 rr = raw[i * NI + ni];
                acc[mb * RB + i][ni] = colv[ni] + bepi_f32x4{xbf16_lo(rr[0]), xbf16_hi(rr[0]), xbf16_lo(rr[1]), xbf16_hi(rr[1])};
            }
        __builtin_amdgcn_sched_barrier(0);      // (one batch's loads at a time)
    }
}

// interior: the whole tile lies inside M x N (uniform)
template <int MI, int NI, int WM, int WN, int RBM>
__device__ __forceinline__ void gemm_acc_rows_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], const bepi_f32x4 (&colv)[NI], const int (&ncol)[NI], const bool with_r,
                                                   const bool with_v, const bool interior, const int m0, const int wave, const int lane, const int HoWo) {
    if constexpr (RBM > 0) {     // (RBM = 0: the kernel-row convolution's NI = 5 instantiations, 248 registers in their k loop, spill 21 with the batched form beside the general one)
        if (with_r && !with_v && interior) {     // the common case (attention / feed-forward output projections, the ResBlocks' second convolution): batched, no predicates
            gemm_acc_resid_bf16<MI, NI, WM, WN, RBM>(p, acc, colv, ncol[0], m0, wave, lane);
            return;
        }
    }
    const int wm = wave / WN;
    const int c15 = lane & 15;
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        const int mm = m < p.M ? m : 0;
        const unsigned short* row = Rh + (long long)mm * p.ldr;
        const float* rv = with_v ? p.rowvec + (long long)(mm / HoWo) * p.rowvec_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = ncol[ni];
            bepi_f32x4 v = colv[ni];
            // (rows / columns past the tile's extent read the zero page: a pointer select, no predicated load)
            if (with_v) v += *reinterpret_cast<const bepi_f32x4*>(n < p.N ? reinterpret_cast<const void*>(rv + n) : p.zero_page);
            if (with_r) {
                const bepi_u32x2 rr = *reinterpret_cast<const bepi_u32x2*>((m < p.M && n < p.N) ? reinterpret_cast<const void*>(row + n) : p.zero_page);
                v += bepi_f32x4{xbf16_lo(rr[0]), xbf16_hi(rr[0]), xbf16_lo(rr[1]), xbf16_hi(rr[1])};
            }
            acc[mi][ni] = v;
        }
    }
}

// The per-column part of that initial value (the bias; zeros without one): NI 16-byte loads.  The tile loop of k_gemm_bf16x.hip requests the NEXT tile's in front of the
// epilogue's stores -- vmcnt retires in order, so a load issued behind 16 stores waits for all of them (profiles/r06zh_*: 2.1 us per tile in front of the k loop).
template <int NI, int WN, int GEGLU_T = -1>
__device__ __forceinline__ void gemm_acc_cols_bf16(const ConvGemm& p, bepi_f32x4 (&colv)[NI], const int n0, const int wave, const int lane) {
    const bool geglu = GEGLU_T < 0 ? p.geglu != 0 : GEGLU_T == 1;
    const bool with_b = (p.resid_acc & 2) && p.bias;
    const int wn = wave % WN;
    const int g4 = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = geglu ? n0 + wn * (8 * NI) + (ni >> 1) * 16 + g4 * 4 : n0 + (wn * NI + ni) * 16 + g4 * 4;
        colv[ni] = bepi_f32x4{0.f, 0.f, 0.f, 0.f};
        if (with_b) colv[ni] = *reinterpret_cast<const bepi_f32x4*>(n < p.N ? reinterpret_cast<const void*>(p.bias + n + ((geglu && (ni & 1)) ? p.N : 0)) : p.zero_page);
    }
}

template <int MI, int NI, int WM, int WN, int GEGLU_T = -1, int RBM = (NI > 4 ? 2 : 4)>
__device__ __forceinline__ void gemm_acc_init_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], const bepi_f32x4 (&colv)[NI], const int m0, const int n0, const int wave,
                                                   const int lane, const int HoWo) {
    const bool geglu = GEGLU_T < 0 ? p.geglu != 0 : GEGLU_T == 1;
    const bool with_r = GEGLU_T != 1 && (p.resid_acc & 1);
    bool with_v = GEGLU_T != 1 && (p.resid_acc & 2) && p.rowvec;
    if (!with_r && !with_v) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = colv[ni];
        return;
    }
    const int wn = wave % WN;
    const int g4 = lane >> 4;
    int ncol[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) ncol[ni] = geglu ? n0 + wn * (8 * NI) + (ni >> 1) * 16 + g4 * 4 : n0 + (wn * NI + ni) * 16 + g4 * 4;
    // a tile whose rows belong to ONE sample (every level but the 8 x 8 one): the time-embedding row is a per-column term like the bias -- NI loads in one batch instead of
    // MI x NI loads, each waited for where it is issued, behind a sample-index division per fragment row (the same sum in the same order: bias + row, then the residual)
    bepi_f32x4 cv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) cv[ni] = colv[ni];
    constexpr bool FOLD = !(GEGLU_T < 0 && MI * NI >= 40);     // (the one-tile 256 x 320 bf16 form, every epilogue decision at run time, spills 4 registers with one more path)
    if (FOLD && with_v) {
        const int mlast = (m0 + 16 * MI * WM <= p.M ? m0 + 16 * MI * WM : p.M) - 1;
        const int s0 = m0 / HoWo;
        if (mlast / HoWo == s0) {
            const float* rv = p.rowvec + (long long)s0 * p.rowvec_stride;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) cv[ni] += *reinterpret_cast<const bepi_f32x4*>(ncol[ni] < p.N ? reinterpret_cast<const void*>(rv + ncol[ni]) : p.zero_page);
            with_v = false;
            if (!with_r) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = cv[ni];
                return;
            }
        }
    }
    gemm_acc_rows_bf16<MI, NI, WM, WN, RBM>(p, acc, cv, ncol, with_r, with_v, m0 + 16 * MI * WM <= p.M && n0 + 16 * NI * WN <= p.N, m0, wave, lane, HoWo);
}

// The one-call forms.  ONE_PASS = false: the two calls above (hipcc then feeds the bias registers to the first matrix instructions instead of copying them into 128 - 160
// accumulators behind a wait: profiles/r06zi_* against r06zl_*, Linear shapes on the 256 x 320 tile -9 % more, MXFP8 image +0.9 %); ONE_PASS = true: round 6's first form in one
// body, which the kernel-row convolution's NI = 5 instantiations need (the two-call form spills 21 - 25 registers there).
template <int MI, int NI, int WM, int WN, int GEGLU_T = -1, bool ONE_PASS = false, int RBM = (NI > 4 ? 2 : 4)>
__device__ __forceinline__ void gemm_acc_init_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], const int m0, const int n0, const int wave, const int lane,
                                                   const int HoWo) {
    if constexpr (!ONE_PASS) {
        bepi_f32x4 colv2[NI];
        gemm_acc_cols_bf16<NI, WN, GEGLU_T>(p, colv2, n0, wave, lane);
        gemm_acc_init_bf16<MI, NI, WM, WN, GEGLU_T, RBM>(p, acc, colv2, m0, n0, wave, lane, HoWo);
        return;
    }
    const bool geglu = GEGLU_T < 0 ? p.geglu != 0 : GEGLU_T == 1;
    const bool with_r = GEGLU_T != 1 && (p.resid_acc & 1);
    const bool with_b = (p.resid_acc & 2) && p.bias;
    bool with_v = GEGLU_T != 1 && (p.resid_acc & 2) && p.rowvec;
    if (!with_r && !with_b && !with_v) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = bepi_f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    // per-column terms first (the same for every fragment row): NI 16-byte loads
    bepi_f32x4 colv[NI];
    int ncol[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = geglu ? n0 + wn * (8 * NI) + (ni >> 1) * 16 + g4 * 4 : n0 + (wn * NI + ni) * 16 + g4 * 4;
        ncol[ni] = n;
        colv[ni] = bepi_f32x4{0.f, 0.f, 0.f, 0.f};
        if (with_b) colv[ni] = *reinterpret_cast<const bepi_f32x4*>(n < p.N ? reinterpret_cast<const void*>(p.bias + n + ((geglu && (ni & 1)) ? p.N : 0)) : p.zero_page);
    }
    if (with_v) {     // a tile inside one sample: the time-embedding row joins the per-column terms (see the two-call form)
        const int mlast = (m0 + 16 * MI * WM <= p.M ? m0 + 16 * MI * WM : p.M) - 1;
        const int s0 = m0 / HoWo;
        if (mlast / HoWo == s0) {
            const float* rv0 = p.rowvec + (long long)s0 * p.rowvec_stride;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) colv[ni] += *reinterpret_cast<const bepi_f32x4*>(ncol[ni] < p.N ? reinterpret_cast<const void*>(rv0 + ncol[ni]) : p.zero_page);
            with_v = false;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        const int mm = m < p.M ? m : 0;
        const unsigned short* row = Rh + (long long)mm * p.ldr;
        const float* rv = with_v ? p.rowvec + (long long)(mm / HoWo) * p.rowvec_stride : nullptr;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = ncol[ni];
            bepi_f32x4 v = colv[ni];
            // (rows / columns past the tile's extent read the zero page: a pointer select, no predicated load)
            if (with_v) v += *reinterpret_cast<const bepi_f32x4*>(n < p.N ? reinterpret_cast<const void*>(rv + n) : p.zero_page);
            if (with_r) {
                const bepi_u32x2 rr = *reinterpret_cast<const bepi_u32x2*>((m < p.M && n < p.N) ? reinterpret_cast<const void*>(row + n) : p.zero_page);
                v += bepi_f32x4{xbf16_lo(rr[0]), xbf16_hi(rr[0]), xbf16_lo(rr[1]), xbf16_hi(rr[1])};
            }
            acc[mi][ni] = v;
        }
    }
}

// pre_synced: the caller has already passed a workgroup barrier behind the last k tile (the persistent kernel, which issues the next tile's first DMA between that
//             barrier and this epilogue) -- smem_x is then the stage the next tile does NOT land in.
// MODE: -1 = every decision below is taken at run time from p (the one-tile-per-workgroup kernels); 0 / 2 = bf16 output through the 16-byte paths, known at compile
//             time to be plain (no residual) / the GEGLU gate (the persistent kernel: inside its tile loop the run-time form with all its paths costs 200 spilled
//             registers, the residual path alone 49) -- the launcher checks what the run-time form checks.
template <int MI, int NI, int WM, int WN, int MODE = -1>
__device__ __forceinline__ void gemm_epilogue_bf16(const ConvGemm& p, bepi_f32x4 (&acc)[MI][NI], unsigned char* smem_x, const int m0, const int n0,
                                                   const int z, const int wave, const int lane, const int HoWo, const bool pre_synced = false) {
    typedef bepi_f32x4 f32x4;
    typedef bepi_u32x4 u32x4;
    typedef bepi_u32x2 u32x2;
    constexpr int WNC = 16 * NI;        // columns of a wave tile
    const bool geglu = MODE < 0 ? p.geglu != 0 : MODE == 2;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int c15 = lane & 15, g4 = lane >> 4;
    // ---- epilogue: fp32 bias + time-embedding row + (bf16) residual, then bf16 or fp32 store ----------
    // A lane holds 4 consecutive channels of 16 different rows, so storing straight from the accumulators issues
    // 8-byte pieces at a row stride (measured: 38k cycles for a 256x320 tile, store-issue bound).  Each wave instead
    // transposes one 16-row fragment group at a time through its own LDS scratch (the stages are free now) and
    // writes whole 160-byte row segments with 16-byte lanes; the residual is read the same way.
    const bool split = MODE < 0 ? p.splits > 1 : false;
    const bool out_f32 = MODE < 0 ? (split || p.out_mode == 1) : false;
    float* Cf = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(p.resid);
    const int ldc = split ? p.N : p.ldc;
    const bool has_resid = MODE < 0 ? (!split && p.resid && !(p.resid_acc & 1)) : false;   // (resid_acc bit 0: already in the accumulators)
    const bool vec_ok = MODE < 0 ? (((p.N & 7) == 0) && ((ldc & 7) == 0) && ((p.ldr & 7) == 0 || !has_resid)) : true;
    constexpr int LDSW = WNC + 4;       // scratch row stride in floats (336 B for NI = 5: 16-byte aligned, rows on distinct banks)
    // ---- the lean form (round 6; ConvGemm::variant bit 3 switches it off): an INTERIOR tile with bf16 output whose bias / time-embedding row / residual are already in the
    // accumulators (resid_acc) needs no bounds check, no global load and no per-row address arithmetic.  The general form below spends ~110 executed instructions per 16-row
    // fragment group on them (an integer division for the sample index of a time-embedding row it does not add, 64-bit row address products, exec-mask branches around
    // loads that are not taken, register copies behind predicated LDS reads that put an lgkmcnt(0) right behind every read) -- profiles/r06zg_*: 4.1 us of a 256 x 256 tile's
    // 20.6 us at K = 320 sat in the epilogue, of which the stores were 1 - 2 us (r06zf_*).  Here every lane-derived address is computed ONCE per tile: one LDS write
    // address, NR LDS read addresses, NR 32-bit store offsets against a wave-uniform row pointer that advances by 16 rows per group.  Same values, same single rounding.
    constexpr int BMT = 16 * MI * WM, BNT = 16 * NI * WN;
    if (!(p.variant & 8) && m0 + BMT <= p.M) {
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        const int wmu = wu / WN, wnu = wu - wmu * WN;
        if (geglu) {
            if constexpr (NI % 2 == 0) {
                if (n0 + BNT / 2 <= p.N && (MODE >= 0 || p.out_mode != 1) && (!p.bias || (p.resid_acc & 2))) {
                    constexpr int WNO = WNC / 2, RSB = WNO * 2 + 16, CH = WNO / 8, NR = (16 * CH + 63) / 64, REM = 16 * CH - (NR - 1) * 64;
                    static_assert(((RSB / 16) & 1) == 1 && ((NR * 64 - 1) / CH) * RSB + CH * 16 <= 16 * (WNO + 4) * 4, "scratch rows on distinct 16-byte bank slots, reads inside the wave's scratch");
                    if (!pre_synced) __syncthreads();
                    unsigned char* sb = smem_x + wu * (16 * (WNO + 4) * 4);
                    unsigned char* wr = sb + c15 * RSB + g4 * 8;
                    const unsigned char* rd[NR];
                    unsigned goff[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int q = r * 64 + lane, row = q / CH, c8 = q - row * CH;
                        rd[r] = sb + row * RSB + c8 * 16;
                        goff[r] = (unsigned)(row * p.ldc + c8 * 8) * 2u;
                    }
                    unsigned char* gbase = reinterpret_cast<unsigned char*>(p.C) + ((long long)(m0 + wmu * MI * 16) * p.ldc + n0 + wnu * WNO) * 2;
                    const long long gstep = (long long)p.ldc * 32;
                    auto stage_g = [&](int mi) {
#pragma unroll
                        for (int j = 0; j < NI / 2; ++j) {
                            const f32x4 v = acc[mi][2 * j], g = acc[mi][2 * j + 1];
                            const kc_f32x2 g01 = gelu_gate_fast2(kc_f32x2{g[0], g[1]}), g23 = gelu_gate_fast2(kc_f32x2{g[2], g[3]});
                            const kc_f32x2 o01 = kc_f32x2{v[0], v[1]} * g01, o23 = kc_f32x2{v[2], v[3]} * g23;
                            *reinterpret_cast<u32x2*>(wr + j * 32) = u32x2{xpack_bf16x2(o01[0], o01[1]), xpack_bf16x2(o23[0], o23[1])};
                        }
                    };
                    stage_g(0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        u32x4 o[NR];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r = 0; r < NR; ++r) o[r] = *reinterpret_cast<const u32x4*>(rd[r]);
                        __builtin_amdgcn_wave_barrier();
                        if (mi + 1 < MI) stage_g(mi + 1);
#pragma unroll
                        for (int r = 0; r < NR; ++r)
                            if (r + 1 < NR || REM == 64 || lane < REM) *reinterpret_cast<u32x4*>(gbase + goff[r]) = o[r];
                        gbase += gstep;
                    }
                    return;
                }
            }
        } else if (n0 + BNT <= p.N && !out_f32 && !has_resid && vec_ok && (!(p.bias || p.rowvec) || (p.resid_acc & 2))) {
            constexpr int RSB = WNC * 2 + 16, CH = WNC / 8, NR = (16 * CH + 63) / 64, REM = 16 * CH - (NR - 1) * 64;
            static_assert(((RSB / 16) & 1) == 1 && ((NR * 64 - 1) / CH) * RSB + CH * 16 <= 16 * LDSW * 4, "scratch rows on distinct 16-byte bank slots, reads inside the wave's scratch");
            if (!pre_synced) __syncthreads();
            unsigned char* sb = smem_x + wu * (16 * LDSW * 4);
            unsigned char* wr = sb + c15 * RSB + g4 * 8;
            const unsigned char* rd[NR];
            unsigned goff[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int q = r * 64 + lane, row = q / CH, c8 = q - row * CH;
                rd[r] = sb + row * RSB + c8 * 16;
                goff[r] = (unsigned)(row * ldc + c8 * 8) * 2u;
            }
            unsigned char* gbase = reinterpret_cast<unsigned char*>(Ch) + ((long long)(m0 + wmu * MI * 16) * ldc + n0 + wnu * WNC) * 2;
            const long long gstep = (long long)ldc * 32;
            auto stage_l = [&](int mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4 v = acc[mi][ni];
                    *reinterpret_cast<u32x2*>(wr + ni * 32) = u32x2{xpack_bf16x2(v[0], v[1]), xpack_bf16x2(v[2], v[3])};
                }
            };
            stage_l(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                u32x4 o[NR];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < NR; ++r) o[r] = *reinterpret_cast<const u32x4*>(rd[r]);      // (a last partial round reads past row 15, inside this wave's scratch; only its store is predicated)
                __builtin_amdgcn_wave_barrier();
                if (mi + 1 < MI) stage_l(mi + 1);    // (in-order LDS: lands behind the reads above)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (r + 1 < NR || REM == 64 || lane < REM) *reinterpret_cast<u32x4*>(gbase + goff[r]) = o[r];
                gbase += gstep;
            }
            return;
        }
    }
    if (geglu) {   // launch-side guarantees: NI even, no split-K, N % 8 == 0, ldc % 8 == 0, no rowvec / residual
        if constexpr (NI % 2 == 0) {
            constexpr int WNO = WNC / 2;     // output columns of a wave tile
            constexpr int LDSW2 = WNO + 4;
            if (!pre_synced) __syncthreads();
            float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW2 * 4));
            const int nw0 = n0 + wn * WNO;
            auto gate = [&](int mi, int j) {
                const int n = nw0 + j * 16 + g4 * 4;
                f32x4 v = acc[mi][2 * j], g = acc[mi][2 * j + 1];
                if (p.bias && !(p.resid_acc & 2) && n < p.N) {     // (resid_acc bit 1: already in the accumulators)
                    v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    g += *reinterpret_cast<const f32x4*>(p.bias + p.N + n);
                }
                // (k_common.hpp: erff() cost 5 us per 256 x 256 tile here, profiles/r05ad_*; round 6: two gates per instruction where the operation has a packed form)
                const kc_f32x2 g01 = gelu_gate_fast2(kc_f32x2{g[0], g[1]}), g23 = gelu_gate_fast2(kc_f32x2{g[2], g[3]});
                const kc_f32x2 o01 = kc_f32x2{v[0], v[1]} * g01, o23 = kc_f32x2{v[2], v[3]} * g23;
                return f32x4{o01[0], o01[1], o23[0], o23[1]};
            };
            constexpr int CH = WNO / 8;
            if (MODE >= 0 || p.out_mode != 1) {
                // bf16 output: rounded before the transpose, 2-byte scratch (see the main path below), the gate of group mi + 1 behind the stores of group mi
                constexpr int RSB = WNO * 2 + 16;
                static_assert(((RSB / 16) & 1) == 1, "scratch rows must start on distinct 16-byte bank slots");
                constexpr int NR = (16 * CH + 63) / 64;
                unsigned char* sb = smem_x + wave * (16 * LDSW2 * 4);
                auto stage_b = [&](int mi) {
#pragma unroll
                    for (int j = 0; j < NI / 2; ++j) {
                        const f32x4 o = gate(mi, j);
                        const u32x2 w = {xpack_bf16x2(o[0], o[1]), xpack_bf16x2(o[2], o[3])};
                        *reinterpret_cast<u32x2*>(sb + c15 * RSB + j * 32 + g4 * 8) = w;
                    }
                };
                stage_b(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int mrow0 = m0 + (wm * MI + mi) * 16;
                    u32x4 o[NR];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int q = r * 64 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        if (q < 16 * CH) o[r] = *reinterpret_cast<const u32x4*>(sb + row * RSB + c8 * 16);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (mi + 1 < MI) stage_b(mi + 1);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int q = r * 64 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        const int m = mrow0 + row, n = nw0 + c8 * 8;
                        if (q < 16 * CH && m < p.M && n < p.N) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.C) + (long long)m * p.ldc + n) = o[r];
                    }
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int mrow0 = m0 + (wm * MI + mi) * 16;
#pragma unroll
                    for (int j = 0; j < NI / 2; ++j) *reinterpret_cast<f32x4*>(scr + c15 * LDSW2 + j * 16 + g4 * 4) = gate(mi, j);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                        const int q = q0 + lane;
                        const int row = q / CH, c8 = q - row * CH;
                        const int m = mrow0 + row, n = nw0 + c8 * 8;
                        if (q < 16 * CH && m < p.M && n < p.N) {
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8);
                            const f32x4 hi = *reinterpret_cast<const f32x4*>(scr + row * LDSW2 + c8 * 8 + 4);
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n) = lo;
                            *reinterpret_cast<f32x4*>(p.C + (long long)m * p.ldc + n + 4) = hi;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        return;
    }
    if (vec_ok) {
        if (!pre_synced) __syncthreads();                // every wave is done with the last k tile
        float* scr = reinterpret_cast<float*>(smem_x + wave * (16 * LDSW * 4));
        const int nw0 = n0 + wn * WNC;
        // fragment group mi (16 rows x WNC columns of this wave) -> scratch, with the per-column / per-sample terms
        auto stage = [&](int mi) {
            const int m = m0 + (wm * MI + mi) * 16 + c15;
            const int smp = (m < p.M ? m : 0) / HoWo;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = nw0 + ni * 16 + g4 * 4;
                f32x4 v = acc[mi][ni];
                if (!split && !(p.resid_acc & 2) && n < p.N) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                }
                *reinterpret_cast<f32x4*>(scr + c15 * LDSW + ni * 16 + g4 * 4) = v;
            }
        };
        if (!out_f32 && !has_resid) {
            // bf16 output without a residual: the values are rounded to bf16 BEFORE the transpose (the same single rounding), so the scratch holds 2 bytes per
            // element -- 5 ds_write_b64 + 3 ds_read_b128 per fragment group where the fp32 scratch below takes 5 ds_write_b128 + 6 ds_read_b128 whose 32-byte lane
            // stride uses half the banks.  Row stride WNC 2 + 16 bytes = 4 x odd dwords: the 16 rows of a half-wave's ds_write_b64 tile the 64 banks.
            constexpr int RSB = WNC * 2 + 16;
            static_assert(((RSB / 16) & 1) == 1, "scratch rows must start on distinct 16-byte bank slots");
            constexpr int CH = WNC / 8;   // 16-byte bf16 chunks per row
            constexpr int NR = (16 * CH + 63) / 64;
            unsigned char* sb = smem_x + wave * (16 * LDSW * 4);
            auto stage_b = [&](int mi) {
                const int m = m0 + (wm * MI + mi) * 16 + c15;
                const int smp = (m < p.M ? m : 0) / HoWo;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = nw0 + ni * 16 + g4 * 4;
                    f32x4 v = acc[mi][ni];
                    if (!(p.resid_acc & 2) && n < p.N) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    }
                    const u32x2 w = {xpack_bf16x2(v[0], v[1]), xpack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(sb + c15 * RSB + ni * 32 + g4 * 8) = w;
                }
            };
            stage_b(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                u32x4 o[NR];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    if (q < 16 * CH) o[r] = *reinterpret_cast<const u32x4*>(sb + row * RSB + c8 * 16);
                }
                __builtin_amdgcn_wave_barrier();
                if (mi + 1 < MI) stage_b(mi + 1);    // (in-order LDS: lands behind the reads above)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o[r];
                }
            }
        } else if (!out_f32) {
            // bf16 output.  Software-pipelined over the fragment groups: group mi is read back row-coalesced into registers, group mi + 1 is staged into the SAME
            // scratch right behind those reads (the LDS executes one wave's instructions in order, so the writes land after them), and only then are the rows of
            // group mi converted and stored -- the LDS round trip of the next group hides behind the conversion and the store issue of this one
            constexpr int CH = WNC / 8;   // 16-byte bf16 chunks per row
            constexpr int NR = (16 * CH + 63) / 64;
            stage(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                f32x4 lo[NR], hi[NR];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    if (q < 16 * CH) {
                        lo[r] = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8);
                        hi[r] = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c8 * 8 + 4);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (mi + 1 < MI) stage(mi + 1);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int q = r * 64 + lane;
                    const int row = q / CH, c8 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c8 * 8;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        f32x4 l = lo[r], h = hi[r];
                        if (has_resid) {
                            const u32x4 rr = *reinterpret_cast<const u32x4*>(Rh + (long long)m * p.ldr + n);
                            l[0] += xbf16_lo(rr[0]); l[1] += xbf16_hi(rr[0]); l[2] += xbf16_lo(rr[1]); l[3] += xbf16_hi(rr[1]);
                            h[0] += xbf16_lo(rr[2]); h[1] += xbf16_hi(rr[2]); h[2] += xbf16_lo(rr[3]); h[3] += xbf16_hi(rr[3]);
                        }
                        const u32x4 o = {xpack_bf16x2(l[0], l[1]), xpack_bf16x2(l[2], l[3]), xpack_bf16x2(h[0], h[1]), xpack_bf16x2(h[2], h[3])};
                        *reinterpret_cast<u32x4*>(Ch + (long long)m * ldc + n) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int mrow0 = m0 + (wm * MI + mi) * 16;
                stage(mi);
                __builtin_amdgcn_wave_barrier();
                constexpr int CH = WNC / 4;   // 16-byte fp32 chunks per row
#pragma unroll
                for (int q0 = 0; q0 < 16 * CH; q0 += 64) {
                    const int q = q0 + lane;
                    const int row = q / CH, c4 = q - row * CH;
                    const int m = mrow0 + row, n = nw0 + c4 * 4;
                    if (q < 16 * CH && m < p.M && n < p.N) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * LDSW + c4 * 4);
                        if (has_resid) {
                            const u32x2 r = *reinterpret_cast<const u32x2*>(Rh + (long long)m * p.ldr + n);
                            v[0] += xbf16_lo(r[0]); v[1] += xbf16_hi(r[0]); v[2] += xbf16_lo(r[1]); v[3] += xbf16_hi(r[1]);
                        }
                        *reinterpret_cast<f32x4*>(Cf + (long long)m * ldc + n) = v;     // (split-K: Cf = this k slice's fp32 slab)
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
    // odd strides / N not a multiple of 8: element-wise stores straight from the accumulators
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
                        if (p.bias && !(p.resid_acc & 2)) sv += p.bias[n + r];
                        if (p.rowvec && !(p.resid_acc & 2)) sv += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                        if (p.resid && !(p.resid_acc & 1)) sv += __uint_as_float((unsigned)Rh[(long long)m * p.ldr + n + r] << 16);
                    }
                    if (out_f32) Cf[(long long)m * ldc + n + r] = sv;
                    else Ch[(long long)m * ldc + n + r] = (unsigned short)xf32_to_bf16_bits(sv);
                }
            }
        }
    }
    }
}

}  // namespace sdmi
