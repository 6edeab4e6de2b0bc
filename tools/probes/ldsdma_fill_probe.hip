// ldsdma_fill_probe.hip -- how fast can ONE compute unit land bytes in its LDS?   (run once at the very end of round 2: profiles/r02_ldsdma_fill_probe.txt)
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_fill_probe.hip -o /tmp/ldsdma_fill_probe && /tmp/ldsdma_fill_probe
//
// Why: in round 2 every LDS-DMA staged GEMM of this repository (k_gemm3x.hip, k_gemm_bf16x.hip, k_fp8.hip) spent the same time per k tile in
// every form of its k loop, and that time is (bytes one k tile stages in LDS) / (10-14 bytes per cycle per CU) -- profiles/README.md, "Where the
// k loop's time goes".  If 10-14 B/clk is what a CU can ingest through `global_load_lds_dwordx4`, the kernels are at their ceiling and the next
// step is a different data path (more reuse per staged byte, or a second ingest path beside the DMA); if a CU can ingest several times that from
// L2, the kernels leave it on the table and the next step is to find out what throttles their fills.  This probe measures the rate directly:
//
//   * one 512-thread workgroup per CU (256 workgroups), an LDS ring of 8 x 16 KiB, W of the 8 waves issue `global_load_lds_dwordx4` (1 KiB per
//     wave-instruction) back to back with `s_waitcnt vmcnt(D)` keeping D instructions per wave in flight; nobody reads the LDS;
//   * source: (0) a private stream per workgroup (HBM), (1) one 2 MiB region per XCD-group read by every workgroup (L2 resident), (2) one
//     64 MiB region read by every workgroup (Infinity Cache resident), (3) 64 KiB per workgroup, read over and over (L2, no sharing);
//   * optionally the other 8 - W waves run dependent-free bf16 matrix instructions (does a busy matrix pipe change the fill rate?);
//   * the same sweep through the REGISTER path (`global_load_dwordx4` -> VGPR -> `ds_write_b128`), the staging the 4-wave kernels use.
// Output per configuration: bytes per cycle per CU (s_memtime over the workgroup's loop) and chip-wide GB/s (hipEvent around the launch).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Args {
    const unsigned char* src;
    unsigned long long region;      // bytes of the region a workgroup walks (power of two)
    unsigned long long wg_stride;   // byte offset between the regions of consecutive workgroups (0: shared)
    int iters;                      // wave-instructions per issuing wave
    int issuers;                    // waves that issue fills (1 .. 8)
    int mfma;                       // 1: the other waves run matrix instructions
    unsigned long long* cycles;     // per workgroup: s_memtime ticks of the fill loop (max over its issuing waves, written by wave 0 after a barrier)
    float* sink;
};

template <int DEPTH, bool REGPATH>
__global__ __launch_bounds__(512) void fill_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 8 slots x 16 KiB
    __shared__ unsigned long long t_wave[8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = a.src + (unsigned long long)blockIdx.x * a.wg_stride;
    const unsigned long long mask = a.region - 1;
    unsigned long long t0 = 0, t1 = 0;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    if (wave < a.issuers) {
        // issuing wave w walks the region with stride issuers * 1 KiB; ring slot = (instruction index) % 16 inside this wave's 16 KiB
        unsigned long long off = (unsigned long long)wave * 1024 + (unsigned long long)lane * 16;
        const unsigned long long step = (unsigned long long)a.issuers * 1024;
        unsigned char* slot0 = smem + wave * 16384;
        t0 = __builtin_readcyclecounter();
        if constexpr (!REGPATH) {
            for (int i = 0; i < a.iters; ++i) {
                __builtin_amdgcn_global_load_lds((global_cvoid*)(base + (off & mask)), (lds_void*)(slot0 + (i & 15) * 1024), 16, 0, 0);
                off += step;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 r[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = u32x4{0, 0, 0, 0};
            for (int i = 0; i < a.iters; i += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    r[d] = *reinterpret_cast<const u32x4*>(base + (off & mask));
                    off += step;
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    asm volatile("" : "+v"(r[d]));      // the load must happen (round 2's first run printed 790 B/clk here: hipcc had removed the dead loads and stores)
                    *reinterpret_cast<volatile u32x4*>(slot0 + ((i + d) & 15) * 1024 + lane * 16) = r[d];
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
    } else if (a.mfma) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
        for (int i = 0; i < a.iters; ++i) {     // about as long as the fill loop of an issuing wave
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, acc3, 0, 0, 0);
        }
    }
    if (lane == 0) t_wave[wave] = t1 - t0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = 0;
        for (int w = 0; w < a.issuers; ++w) m = t_wave[w] > m ? t_wave[w] : m;
        a.cycles[blockIdx.x] = m;
    }
    if (a.sink && acc0[0] + acc1[0] + acc2[0] + acc3[0] == 123.456f) a.sink[threadIdx.x] = smem[threadIdx.x];   // keep everything alive
}

template <int DEPTH, bool REGPATH>
static void run(const char* what, const unsigned char* src, unsigned long long region, unsigned long long wg_stride, int issuers, int mfma,
                unsigned long long* d_cycles, int n_wg, int iters) {
    Args a{src, region, wg_stride, iters, issuers, mfma, d_cycles, nullptr};
    auto k = fill_kernel<DEPTH, REGPATH>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best_ms = 1e30f;
    std::vector<unsigned long long> cyc(n_wg);
    double cyc_mean = 0.0;
    for (int rep = 0; rep < 4; ++rep) {      // the first repetition warms the caches the source is meant to sit in
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(512), 131072, 0, a);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best_ms) {
            best_ms = ms;
            CHECK(hipMemcpy(cyc.data(), d_cycles, n_wg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            cyc_mean = 0.0;
            for (int i = 0; i < n_wg; ++i) cyc_mean += (double)cyc[i];
            cyc_mean /= n_wg;
        }
    }
    const double bytes_wg = (double)issuers * iters * 1024.0;
    std::printf("%-9s %-34s issuers %d depth %2d mfma %d : %6.2f B/clk/CU  (%7.1f GB/s chip, %6.1f GB/s per CU, %.0f us)\n", REGPATH ? "registers" : "lds-dma", what,
                issuers, DEPTH, mfma, bytes_wg / cyc_mean, bytes_wg * n_wg / (best_ms * 1e-3) / 1e9, bytes_wg / (best_ms * 1e-3) / 1e9, best_ms * 1e3);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main() {
    const int n_wg = 256, iters = 4096;                          // 4 MiB per issuing wave
    const unsigned long long stream_per_wg = 32ull << 20;        // private 32 MiB per workgroup: 8 GiB in all (HBM)
    unsigned char* buf;
    unsigned long long* d_cycles;
    CHECK(hipMalloc(&buf, stream_per_wg * n_wg));
    CHECK(hipMemset(buf, 1, stream_per_wg * n_wg));
    CHECK(hipMalloc(&d_cycles, n_wg * sizeof(unsigned long long)));
    struct Src { const char* name; unsigned long long region, stride; };
    const Src srcs[] = {{"HBM stream (32 MiB / workgroup)", stream_per_wg, stream_per_wg},
                        {"2 MiB shared by all (L2)", 2ull << 20, 0},
                        {"64 MiB shared by all (Infinity Cache)", 64ull << 20, 0},
                        {"64 KiB / workgroup, re-read (L2)", 64ull << 10, 64ull << 10}};
    for (const Src& s : srcs)
        for (int issuers : {1, 2, 4, 8})
            for (int mfma : {0, 1}) {
                if (mfma && issuers == 8) continue;
                run<8, false>(s.name, buf, s.region, s.stride, issuers, mfma, d_cycles, n_wg, iters);
            }
    for (const Src& s : srcs) {          // more fills in flight per wave
        run<16, false>(s.name, buf, s.region, s.stride, 8, 0, d_cycles, n_wg, iters);
        run<2, false>(s.name, buf, s.region, s.stride, 8, 0, d_cycles, n_wg, iters);
    }
    for (const Src& s : srcs)
        for (int issuers : {4, 8}) run<8, true>(s.name, buf, s.region, s.stride, issuers, 0, d_cycles, n_wg, iters);
    CHECK(hipFree(buf));
    CHECK(hipFree(d_cycles));
    return 0;
}
