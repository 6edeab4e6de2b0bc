// mx_probe.hip -- pins down the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, E8M0 block scales) on
// gfx950 before k_gemm_fp8x.hip relies on it.  No ISA document is available offline, so three lane->k hypotheses are
// tried against a host fp64 reference; the one that matches is printed.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mx_probe.hip -o stable_diffusion_burn_amd/build/mx_probe && ./.../mx_probe
//
//   D[i][j] = sum_k A[i][k] 2^(sa[i][k/32]-127) * B[k][j] 2^(sb[j][k/32]-127),   i, j < 16, k < 128
//   lane l supplies A row i = l % 16 and B column j = l % 16; which 32 k values, and which scale, is the question.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_once(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, /*cbsz e4m3*/ 0, /*blgp e4m3*/ 0, 0, sa[l], 0, sb[l]);
    d[l] = acc;
}

// scale byte 1 of the VGPR via op_sel
__global__ void mfma_opsel1(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 1, sa[l], 1, sb[l]);
    d[l] = acc;
}

__global__ void cvt_probe(const float* x, unsigned* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], w, false);
        y[i] = (unsigned)w;
    }
}

static double e4m3_value(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    double x = e == 0 ? std::ldexp((double)m, -9) : std::ldexp(1.0 + m / 8.0, e - 7);
    return s ? -x : x;
}

static int k_of(int hyp, int g, int r, int b) {
    switch (hyp) {
        case 0: return 32 * g + 4 * r + b;                               // 32 consecutive k per lane
        case 1: return 64 * (r / 4) + 16 * g + 4 * (r % 4) + b;          // two runs of 16
        default: return 32 * (r / 2) + 8 * g + 4 * (r % 2) + b;          // four runs of 8
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main() {
    std::srand(7);
    unsigned char A[16][128], B[128][16], SA[16][4], SB[16][4];
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 128; ++k) {
            // finite e4m3 codes only (0x7f / 0xff are NaN)
            do { A[i][k] = (unsigned char)(std::rand() & 0xff); } while ((A[i][k] & 0x7f) == 0x7f);
            do { B[k][i] = (unsigned char)(std::rand() & 0xff); } while ((B[k][i] & 0x7f) == 0x7f);
        }
    for (int i = 0; i < 16; ++i)
        for (int q = 0; q < 4; ++q) { SA[i][q] = (unsigned char)(120 + std::rand() % 12); SB[i][q] = (unsigned char)(122 + std::rand() % 10); }
    double ref[16][16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k)
                s += e4m3_value(A[i][k]) * std::ldexp(1.0, SA[i][k / 32] - 127) * e4m3_value(B[k][j]) * std::ldexp(1.0, SB[j][k / 32] - 127);
            ref[i][j] = s;
        }
    i32x8 *da, *db; int *dsa, *dsb; f32x4* dd;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 16));
    int verdict = -1;
    for (int hyp = 0; hyp < 3; ++hyp) {
        for (int opsel = 0; opsel < 2; ++opsel) {
            unsigned char pa[64][32], pb[64][32];
            int sa[64], sb[64];
            for (int l = 0; l < 64; ++l) {
                const int idx = l % 16, g = l / 16;
                for (int r = 0; r < 8; ++r)
                    for (int b = 0; b < 4; ++b) {
                        const int k = k_of(hyp, g, r, b);
                        pa[l][4 * r + b] = A[idx][k];
                        pb[l][4 * r + b] = B[k][idx];
                    }
                // hypothesis for the scale: the lane's value scales the 32-block its FIRST element lies in
                const int blk = k_of(hyp, g, 0, 0) / 32;
                const int other = 0x7f7f7f7f;   // decoys in the unselected bytes (2^0)
                sa[l] = opsel ? ((other & ~0xff00) | (SA[idx][blk] << 8)) : ((other & ~0xff) | SA[idx][blk]);
                sb[l] = opsel ? ((other & ~0xff00) | (SB[idx][blk] << 8)) : ((other & ~0xff) | SB[idx][blk]);
            }
            CK(hipMemcpy(da, pa, sizeof pa, hipMemcpyHostToDevice)); CK(hipMemcpy(db, pb, sizeof pb, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsa, sa, sizeof sa, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, sizeof sb, hipMemcpyHostToDevice));
            if (opsel) hipLaunchKernelGGL(mfma_opsel1, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            else hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            CK(hipDeviceSynchronize());
            float out[64][4];
            CK(hipMemcpy(out, dd, sizeof out, hipMemcpyDeviceToHost));
            // D layout of every 16x16 MFMA on gfx950: lane l, reg r -> row 4*(l/16)+r, col l%16
            double err = 0, mag = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const double want = ref[4 * (l / 16) + r][l % 16];
                    err = std::fmax(err, std::fabs(out[l][r] - want));
                    mag = std::fmax(mag, std::fabs(want));
                }
            std::printf("hypothesis %d (%s), scale byte %d: max|d| = %.3e (|ref|max %.3e) %s\n", hyp,
                        hyp == 0 ? "lane holds k = 32g .. 32g+31" : hyp == 1 ? "two runs of 16" : "four runs of 8", opsel, err, mag,
                        err < 1e-4 * mag ? "MATCH" : "");
            if (err < 1e-4 * mag && opsel == 0 && verdict < 0) verdict = hyp;
        }
    }
    // fp32 -> e4m3 conversion: rounding mode and what happens above 448
    const float xs[16] = {0.0f, 1.0f, 1.0625f, 1.1875f, 448.0f, 464.0f, 480.0f, 1000.0f, -0.001953125f, 0.0009765625f, 0.017f, 3.3f, -449.0f, 1e-9f, 240.0f, 2.5f};
    float* dx; unsigned* dy;
    CK(hipMalloc(&dx, sizeof xs)); CK(hipMalloc(&dy, 8 * 4));
    CK(hipMemcpy(dx, xs, sizeof xs, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dy, 8);
    CK(hipDeviceSynchronize());
    unsigned ys[8];
    CK(hipMemcpy(ys, dy, sizeof ys, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) {
        const unsigned char c = (unsigned char)(ys[i / 2] >> (8 * (i % 2)));
        std::printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", xs[i], c, (c & 0x7f) == 0x7f ? NAN : e4m3_value(c));
    }
    std::printf("RESULT hypothesis=%d\n", verdict);
    return verdict == 0 ? 0 : 1;
}
