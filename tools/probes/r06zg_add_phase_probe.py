"""Temporary instrumentation (NOT part of the product): stamps the phases of the persistent bf16 tile loop (k_gemm_bf16x.hip) with the shader clock and prints their shares from
Engine::bench_conv under option gemm_probe.  Run it on a clean tree, build, run tools/probes/r06zg_probe.py through gpurun, then `git checkout stable_diffusion_burn_amd/csrc`.
Wave 0 of every workgroup accumulates, per tile: acc_init | wait at the first k tile's barrier | k loop | barrier behind the k loop | issue_first | epilogue | setup | loop top."""
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]


def rep(s, old, new, cnt=1):
    assert s.count(old) == cnt, (old, s.count(old))
    return s.replace(old, new)


p = ROOT / "stable_diffusion_burn_amd/csrc/k_gemm_bf16x.hip"
s = p.read_text()
s = rep(s, """    setup();
    int s0 = 0;      // LDS stage of the tile's first k tile
    issue(0);
""", """    unsigned acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tprev = 0;
    unsigned long long rt0 = 0;
    const bool probing = p.probe != nullptr;
#define STAMP(i) do { if (probing) { __builtin_amdgcn_sched_barrier(0); const unsigned tn_ = (unsigned)__builtin_readcyclecounter(); acc_t[i] += tn_ - tprev; tprev = tn_; __builtin_amdgcn_sched_barrier(0); } } while (0)
    if (probing) { rt0 = __builtin_amdgcn_s_memrealtime(); tprev = (unsigned)__builtin_readcyclecounter(); }
    setup();
    int s0 = 0;      // LDS stage of the tile's first k tile
    issue(0);
""")
s = rep(s, """    for (;;) {
    // the bias, or zero; + the residual tile (ConvGemm::resid_acc)
""", """    for (;;) {
    STAMP(7);
    // the bias, or zero; + the residual tile (ConvGemm::resid_acc)
""")
s = rep(s, """    for (int t = 0; t < n_t; ++t) {
        const int cur = (s0 + t) & 1;
        sdmi_dma_landed(); """, """    STAMP(0);
    for (int t = 0; t < n_t; ++t) {
        const int cur = (s0 + t) & 1;
        sdmi_dma_landed(); """)
s = rep(s, """        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
""", """        __syncthreads();                    // k tile t is in LDS; every wave is done with stage cur ^ 1
        if (t == 0) STAMP(1);
""")
s = rep(s, """    if constexpr (!PERSIST) {
        gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_x, m0, n0, z, wave, lane, HoWo);
        break;
    } else {
""", """    STAMP(2);
    if constexpr (!PERSIST) {
        gemm_epilogue_bf16<MI, NI, WM, WN>(p, acc, smem_x, m0, n0, z, wave, lane, HoWo);
        STAMP(5);
        if (probing) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        STAMP(6);      // (one-tile form: the drain of its own stores)
        break;
    } else {
""")
s = rep(s, """        __syncthreads();                      // every wave is done with stage L (and, since the top of the last k iteration, with stage L ^ 1)
        if (more) {""", """        __syncthreads();                      // every wave is done with stage L (and, since the top of the last k iteration, with stage L ^ 1)
        STAMP(3);
        if (more) {""")
s = rep(s, """        // (the lane index is made opaque per tile: the epilogue's lane-derived offsets""", """        STAMP(4);
        // (the lane index is made opaque per tile: the epilogue's lane-derived offsets""")
s = rep(s, """        if (!more) break;
        s0 = L;""", """        STAMP(5);
        if (!more) break;
        s0 = L;""")
s = rep(s, """        advance_k();      // (k tile 0 is on its way)
    }
    }
""", """        advance_k();      // (k tile 0 is on its way)
        STAMP(6);
    }
    }
    if (probing && wave == 0 && lane == 0) {
        unsigned long long* d = p.probe + 24ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.z);
        for (int i = 0; i < 8; ++i) d[i] = acc_t[i];
        d[8] = __builtin_amdgcn_s_memrealtime() - rt0;
        d[9] = 1;
    }
""")
p.write_text(s)

p = ROOT / "stable_diffusion_burn_amd/csrc/engine.cpp"
s = p.read_text()
s = rep(s, "        if (opt_gemm_probe_ && a.p3) {", "        if (opt_gemm_probe_) {")
s = rep(s, """            if (opt_bench_cold_) {
                Buf flush(this, (size_t)512 << 20);""", """            if (opt_bench_cold_ && a.p3) {
                Buf flush(this, (size_t)512 << 20);""")
s = rep(s, """            probe_report(pb.p, kMaxBlocks, n, cin, h, w, cout, k, tile_cfg, splitk);""", """            if (a.p3) probe_report(pb.p, kMaxBlocks, n, cin, h, w, cout, k, tile_cfg, splitk);
            else {
                std::vector<unsigned long long> hb(kMaxBlocks * 24);
                SDMI_HIP(hipMemcpyAsync(hb.data(), pb.p, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
                SDMI_HIP(hipStreamSynchronize(stream_));
                std::vector<double> v[10];
                for (size_t b = 0; b < kMaxBlocks; ++b) { const unsigned long long* d = &hb[24 * b]; if (!d[9]) continue; for (int i = 0; i < 9; ++i) v[i].push_back((double)d[i]); }
                auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x.empty() ? 0.0 : x[x.size() / 2]; };
                double tot = 0; for (int i = 0; i < 8; ++i) tot += med(v[i]);
                std::fprintf(stderr, "bf16x_probe cin=%d cout=%d tile=%d: %zu wgs, realtime %.2f us; shares: acc_init %.3f first_wait %.3f kloop %.3f post_barrier %.3f issue_first %.3f epilogue %.3f setup %.3f looptop %.3f (cycles total %.0f)\\n",
                             cin, cout, tile_cfg, v[0].size(), med(v[8]) * 0.01, med(v[0]) / tot, med(v[1]) / tot, med(v[2]) / tot, med(v[3]) / tot, med(v[4]) / tot, med(v[5]) / tot, med(v[6]) / tot, med(v[7]) / tot, tot);
            }""")
p.write_text(s)
print("instrumented")
