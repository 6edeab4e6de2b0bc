#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, unsigned* __restrict__ out, const unsigned* zero) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gvoid;
    // each wave copies 1 KiB: lane i loads 16 B from a permuted source position
    const unsigned* g = (lane == 5) ? zero : src + (wave * 64 + (lane ^ 3)) * 4;
    lds_void* dst = (lds_void*)(smem + wave * 1024);
    __builtin_amdgcn_global_load_lds((gvoid*)g, dst, 16, 0, 0);
    __syncthreads();
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + tid * 16);
    *reinterpret_cast<u32x4*>(out + tid * 4) = v;
}
int main() {
    unsigned *src, *out, *zero;
    hipMalloc(&src, 4096); hipMalloc(&out, 4096); hipMalloc(&zero, 256);
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice); hipMemset(zero, 0, 256); hipMemset(out, 0xff, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, src, out, zero);
    unsigned o[1024]; hipMemcpy(o, out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) { int lane = t & 63, w = t >> 6; unsigned exp0 = (lane == 5) ? 0 : (unsigned)((w * 64 + (lane ^ 3)) * 4);
        for (int j = 0; j < 4; ++j) { unsigned e = (lane == 5) ? 0 : exp0 + j; if (o[t * 4 + j] != e) { if (bad < 5) printf("t %d j %d got %u exp %u\n", t, j, o[t*4+j], e); ++bad; } } }
    printf("glds probe: bad = %d\n", bad);
    return bad != 0;
}
