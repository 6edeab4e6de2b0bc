// k_gemm2.hip -- second-generation implicit-GEMM conv / linear kernel (fp32 MFMA).
//
// Math, operand orientation, k order and tile mapping: the notes at the top of k_gemm.hip.
// The memory pipeline (driven by the ISA and measurements of the round-1 first version):
//  * global -> register staging uses RAW BUFFER loads (buffer_load_dwordx4 ... offen)
//    with the hardware range check standing in for every predicate: padding taps,
//    rows beyond M, weight rows beyond N and k beyond K get an out-of-range offset and
//    come back as zeros.  No exec-mask branches, 32-bit offsets, ~8 VALU per load
//    instead of ~20 + a branch: the whole k-loop body is ONE basic block.
//  * LDS tiles are unpadded [rows][32] with the 16-byte chunk index XOR-swizzled by
//    (row & 7): ds_read_b128 fragment reads and ds_write_b128 staging writes are both
//    bank-conflict free (v1's +4 padding was 2-way conflicted for b128 lane groups).
//  * software pipeline inside the k loop: fragments are double buffered across the two
//    16-wide k chunks, the staging registers are written to LDS in the middle of the
//    MFMA stream and immediately re-used for the loads of tile t+2 (a full iteration of
//    latency cover), and the single barrier per k tile sits between two MFMA groups so
//    the first fragment read of the next tile is hidden too.
#include "kernels.hpp"
#include "k_common.hpp"

namespace sdmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOob = 0xFFFFFFF0u;  // byte offset beyond any buffer (< 4 GiB - 16)

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

template <int MI, int NI, int WM, int WN, bool GENERIC>
__global__ __launch_bounds__(256) void conv_gemm2_kernel(const ConvGemm p) {
    constexpr int BM = 16 * MI * WM;
    constexpr int BN = 16 * NI * WN;
    constexpr int PA = (BM + 31) / 32;
    constexpr int PB = (BN + 31) / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BMR = PA * 32;      // LDS rows (>= BM; rows beyond BM/BN are written but never read)
    constexpr int BNR = PB * 32;
    float* As = smem;                 // [2][BMR][32]
    float* Bs = smem + 2 * BMR * 32;  // [2][BNR][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;

    const int MT = (p.M + BM - 1) / BM;
    const int NT = (p.N + BN - 1) / BN;
    const GemmWork gw = gemm_work_of_block(p, MT, NT);
    if (!gw.live) return;
    const int lid = gw.lid;
    const int tm = gw.tm;
    const int tn = gw.tn;
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int z = gw.z;
    const int kt_begin = z * p.kt_per_split;
    const int kt_end = min(kt_begin + p.kt_per_split, p.kt_total);
    const int n_t = kt_end - kt_begin;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bt), 0, (int)p.b_bytes, 0x00020000);

    const int lrow = tid >> 3;
    const int kq = tid & 7;
    const int T = p.KH * p.KW;
    const int HoWo = p.Ho * p.Wo;
    const int Hin = p.Hs << p.ups;
    const int Win = p.Ws << p.ups;
    const unsigned pix_bytes = (unsigned)p.a_ld * 4u;

    int a_iy0[PA], a_ix0[PA];
    unsigned a_nboff[PA];
    bool a_ok[PA];
#pragma unroll
    for (int pa = 0; pa < PA; ++pa) {
        const int r = pa * 32 + lrow;
        const int m = m0 + r;
        const bool ok = (r < BM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int nb = mm / HoWo;
        const int rem = mm - nb * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        a_ok[pa] = ok;
        a_nboff[pa] = (unsigned)nb * (unsigned)(p.Hs * p.Ws) * pix_bytes;
        a_iy0[pa] = oy * p.stride - p.pad;
        a_ix0[pa] = ox * p.stride - p.pad;
    }
    unsigned b_off[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int r = pb * 32 + lrow;
        const int n = n0 + r;
        const bool ok = (r < BN) && (n < p.N);
        b_off[pb] = ok ? ((unsigned)n * (unsigned)p.b_ld * 4u + (unsigned)kq * 16u) : kOob;
    }

    int cs = kt_begin / T;
    int tap0 = kt_begin - cs * T;
    int ky = tap0 / p.KW;
    int kx = tap0 - ky * p.KW;
    int kt_next = kt_begin;  // next k tile gload() will fetch

    f32x4 ra[PA], rb[PB];

    auto gload = [&]() {
        const bool tile_ok = kt_next < kt_end;
        const unsigned k0b = (unsigned)kt_next * 128u;
        if constexpr (!GENERIC) {
            const unsigned c0b = (unsigned)(cs * 32 + kq * 4) * 4u;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                const int iy = a_iy0[pa] + ky;
                const int ix = a_ix0[pa] + kx;
                const bool ok = tile_ok & a_ok[pa] & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const unsigned pix = (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups));
                const unsigned off = a_nboff[pa] + pix * pix_bytes + c0b;
                ra[pa] = buf_load16(rsA, ok ? off : kOob);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const bool ok = tile_ok & (b_off[pb] != kOob);
                rb[pb] = buf_load16(rsB, ok ? b_off[pb] + k0b : kOob);
            }
            const bool wrap_x = (kx + 1 == p.KW);
            const bool wrap_y = wrap_x && (ky + 1 == p.KH);
            kx = wrap_x ? 0 : kx + 1;
            ky = wrap_x ? (wrap_y ? 0 : ky + 1) : ky;
            cs = wrap_y ? cs + 1 : cs;
        } else {
            const int k = kt_next * 32 + kq * 4;
            const bool kok = tile_ok && (k < p.K);
            const int kk = kok ? k : 0;
            const int sl = kk / p.CS;
            const int ci = kk - sl * p.CS;
            const int gcs = sl / T;
            const int gtap = sl - gcs * T;
            const int gky = gtap / p.KW;
            const int gkx = gtap - gky * p.KW;
            const unsigned c0b = (unsigned)(gcs * p.CS + ci) * 4u;
#pragma unroll
            for (int pa = 0; pa < PA; ++pa) {
                const int iy = a_iy0[pa] + gky;
                const int ix = a_ix0[pa] + gkx;
                const bool ok = kok & a_ok[pa] & ((unsigned)iy < (unsigned)Hin) & ((unsigned)ix < (unsigned)Win);
                const unsigned pix = (unsigned)((iy >> p.ups) * p.Ws + (ix >> p.ups));
                const unsigned off = a_nboff[pa] + pix * pix_bytes + c0b;
                ra[pa] = buf_load16(rsA, ok ? off : kOob);
            }
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const bool ok = kok & (b_off[pb] != kOob);
                rb[pb] = buf_load16(rsB, ok ? b_off[pb] + k0b : kOob);
            }
        }
        ++kt_next;
    };

    // LDS addressing: row r, 16-byte chunk q -> float offset r*32 + ((q ^ (r & 7)) << 2)
    const int st_chunk = (kq ^ (lrow & 7)) << 2;  // rows pa*32 + lrow: (row & 7) == (lrow & 7)
    auto lstore = [&](int buf) {
#pragma unroll
        for (int pa = 0; pa < PA; ++pa) {
            const int r = pa * 32 + lrow;
            *reinterpret_cast<f32x4*>(As + (buf * BMR + r) * 32 + st_chunk) = ra[pa];
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int r = pb * 32 + lrow;
            *reinterpret_cast<f32x4*>(Bs + (buf * BNR + r) * 32 + st_chunk) = rb[pb];
        }
    };

    // fragment reads: lane (c = lane&15, g = lane>>4) reads row base+c, chunk kk*4+g (swizzled by row&7 = c&7,
    // tile row bases are multiples of 16)
    const int c15 = lane & 15, g4 = lane >> 4;
    const int fr_chunk0 = ((0 + g4) ^ (c15 & 7)) << 2;
    const int fr_chunk1 = ((4 + g4) ^ (c15 & 7)) << 2;
    const float* a_row = As + (wm * 16 * MI + c15) * 32;
    const float* b_row = Bs + (wn * 16 * NI + c15) * 32;
    auto lread = [&](int buf, int kk, f32x4 (&a)[MI], f32x4 (&b)[NI]) {
        const int ch = kk ? fr_chunk1 : fr_chunk0;
        const float* ab = a_row + buf * BMR * 32 + ch;
        const float* bb = b_row + buf * BNR * 32 + ch;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(ab + mi * 16 * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bb + ni * 16 * 32);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const f32x4 (&a)[MI], const f32x4 (&b)[NI], int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni][j], a[mi][j], acc[mi][ni], 0, 0, 0);
    };

    f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];

    gload();        // tile 0
    lstore(0);
    gload();        // tile 1 (or zeros)
    __syncthreads();
    lread(0, 0, fa0, fb0);

    for (int t = 0; t < n_t; ++t) {
        const int cur = t & 1;
        lread(cur, 1, fa1, fb1);
        mma(fa0, fb0, 0, 4);
        lstore(cur ^ 1);   // tile t+1 (zeros past the end)
        gload();           // tile t+2 into the just-freed staging registers
        mma(fa1, fb1, 0, 2);
        __syncthreads();
        lread(cur ^ 1, 0, fa0, fb0);
        mma(fa1, fb1, 2, 4);
    }

    // ---- epilogue: a k slice stores its raw partial tile; otherwise bias + time-embedding row + residual -------------
    const bool split = p.splits > 1;
    float* Cbase = split ? (p.slabs + (long long)z * p.slab_stride) : p.C;
    const int ldc = split ? p.N : p.ldc;
    const bool vec_ok = ((p.N & 3) == 0) && ((ldc & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + (wm * MI + mi) * 16 + c15;
        if (m >= p.M) continue;
        const int smp = m / HoWo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 16 + g4 * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[mi][ni];
            if (vec_ok) {
                if (!split) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.rowvec) v += *reinterpret_cast<const f32x4*>(p.rowvec + (long long)smp * p.rowvec_stride + n);
                    if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (long long)m * p.ldr + n);
                }
                if (p.out_mode == 2 && !split) {  // bf16 consumer (the Cin = 4 layers of the bf16 path)
                    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C) + (long long)m * ldc + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned u = __float_as_uint(v[r]);
                        u += 0x7FFFu + ((u >> 16) & 1u);
                        Ch[r] = (unsigned short)(u >> 16);
                    }
                } else if (split) {
                    *reinterpret_cast<f32x4*>(Cbase + (long long)m * ldc + n) = v;     // (Cbase = this k slice's slab)
                } else {
                    *reinterpret_cast<f32x4*>(Cbase + (long long)m * ldc + n) = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < p.N) {
                        float s = v[r];
                        if (!split) {
                            if (p.bias) s += p.bias[n + r];
                            if (p.rowvec) s += p.rowvec[(long long)smp * p.rowvec_stride + n + r];
                            if (p.resid) s += p.resid[(long long)m * p.ldr + n + r];
                        }
                        Cbase[(long long)m * ldc + n + r] = s;
                    }
                }
            }
        }
    }
}

size_t gemm2_tile_lds_bytes(int cfg) {
    const int bmr = (gemm_tile_info(cfg).bm + 31) / 32 * 32, bnr = (gemm_tile_info(cfg).bn + 31) / 32 * 32;
    return (size_t)2 * (bmr + bnr) * 32 * sizeof(float);
}

template <int MI, int NI, int WM, int WN>
static hipError_t launch_cfg2(const ConvGemm& p, size_t lds, dim3 grid, hipStream_t stream) {
    const bool generic = (p.Cin % 32) != 0;
    if (generic) {
        auto k = conv_gemm2_kernel<MI, NI, WM, WN, true>;
        if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    } else {
        auto k = conv_gemm2_kernel<MI, NI, WM, WN, false>;
        if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(k), (int)lds); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_conv_gemm2(const ConvGemm& p, int cfg, hipStream_t stream) {
    if (cfg < 0 || cfg >= kNumGemmTiles) return hipErrorInvalidValue;
    const int bm = gemm_tile_info(cfg).bm, bn = gemm_tile_info(cfg).bn;
    const int MT = (p.M + bm - 1) / bm, NT = (p.N + bn - 1) / bn;
    const int tiles = MT * NT;
    const dim3 grid = gemm_grid(p, tiles);
    const size_t lds = gemm2_tile_lds_bytes(cfg);
    switch (cfg) {
        case 0: return launch_cfg2<4, 4, 2, 2>(p, lds, grid, stream);
        case 1: return launch_cfg2<4, 2, 2, 2>(p, lds, grid, stream);
        case 2: return launch_cfg2<2, 2, 2, 2>(p, lds, grid, stream);
        case 3: return launch_cfg2<8, 4, 2, 2>(p, lds, grid, stream);
        case 4: return launch_cfg2<2, 5, 4, 1>(p, lds, grid, stream);
        case 5: return launch_cfg2<4, 5, 4, 1>(p, lds, grid, stream);
        case 6: return launch_cfg2<2, 4, 2, 2>(p, lds, grid, stream);
        case 7: return launch_cfg2<4, 5, 2, 2>(p, lds, grid, stream);
        case 8: return launch_cfg2<1, 5, 4, 1>(p, lds, grid, stream);
        case 9: return launch_cfg2<2, 5, 2, 2>(p, lds, grid, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace sdmi
