// mx_probe2.hip -- step-by-step decoding of v_mfma_scale_f32_16x16x128_f8f6f4 (see mx_probe.hip): which lanes / registers /
// scale bytes feed which D element.  Prints small tables; read by a human once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void mf(const i32x8* a, const i32x8* b, const int* sa, const int* sb, f32x4* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
    d[l] = acc;
}
static unsigned char pa[64][32], pb[64][32];
static int sa[64], sb[64];
static float out[64][4];
static i32x8 *da, *db; static int *dsa, *dsb; static f32x4* dd;
static void run() {
    hipMemcpy(da, pa, sizeof pa, hipMemcpyHostToDevice); hipMemcpy(db, pb, sizeof pb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, sizeof sa, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, sizeof sb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mf, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    hipDeviceSynchronize();
    hipMemcpy(out, dd, sizeof out, hipMemcpyDeviceToHost);
}
static void ones() { memset(pa, 0x38, sizeof pa); memset(pb, 0x38, sizeof pb); for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 0x7f7f7f7f; }
static void show(const char* what) {
    std::printf("%s\n", what);
    for (int l = 0; l < 64; l += 1) { if (l % 16 == 0) std::printf("  lanes %2d..: ", l); std::printf("[%g %g %g %g] ", out[l][0], out[l][1], out[l][2], out[l][3]); if (l % 16 == 15) std::printf("\n"); }
}
int main() {
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024);
    ones(); run(); std::printf("E1 all ones, scales 2^0: D[0][0]=%g (expect 128)\n", out[0][0]);
    // which scale lane governs a given A register: A = lane 3 reg r only; scale_a doubled in lanes of group q
    for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 8; r += 1) {
        std::printf("E7 A = lane %d (row 3, g=%d), reg %d only; scale_a doubled in lane group q:", 3 + 16 * g, g, r);
        for (int q = 0; q < 4; ++q) {
            ones(); memset(pa, 0, sizeof pa); memset(&pa[3 + 16 * g][4 * r], 0x38, 4);
            for (int l = 0; l < 64; ++l) if (l / 16 == q) sa[l] = 0x7f7f7f80;
            run(); std::printf("  q=%d -> %g", q, out[0][3]);
        }
        std::printf("\n");
    }
    for (int g = 0; g < 4; ++g)
    for (int r = 0; r < 8; r += 4) {
        std::printf("E7b B = lane %d (col 3, g=%d), reg %d only; scale_b doubled in lane group q:", 3 + 16 * g, g, r);
        for (int q = 0; q < 4; ++q) {
            ones(); memset(pb, 0, sizeof pb); memset(&pb[3 + 16 * g][4 * r], 0x38, 4);
            for (int l = 0; l < 64; ++l) if (l / 16 == q) sb[l] = 0x7f7f7f80;
            run(); std::printf("  q=%d -> %g", q, out[3][0]);
        }
        std::printf("\n");
    }
    return 0;
}
