// engine.cpp -- see engine.hpp.  Reference citations are relative to
// Gadersd/stable-diffusion-burn (src/model/...).
#include "engine.hpp"
#include "mpk_reader.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace sdmi {

// =============================================================================
// DevPool
// =============================================================================
static constexpr size_t kAlign = 256;
static constexpr size_t kSlabMin = (size_t)1 << 30;  // 1 GiB

DevPool::~DevPool() {
    for (auto& s : slabs_) (void)hipFree(s.base);
}

void* DevPool::alloc(size_t bytes) {
    if (bytes == 0) bytes = kAlign;
    bytes = (bytes + kAlign - 1) / kAlign * kAlign;
    for (int pass = 0; pass < 2; ++pass) {
        for (size_t si = 0; si < slabs_.size(); ++si) {
            auto& fl = slabs_[si].free_list;
            for (size_t i = 0; i < fl.size(); ++i) {
                if (fl[i].size >= bytes) {
                    void* p = slabs_[si].base + fl[i].off;
                    if (fl[i].size == bytes) fl.erase(fl.begin() + i);
                    else { fl[i].off += bytes; fl[i].size -= bytes; }
                    live_[p] = Live{(int)si, bytes, ++serial_};
                    in_use_ += bytes;
                    high_ = std::max(high_, in_use_);
                    return p;
                }
            }
        }
        // no fit: new slab
        size_t sz = std::max(bytes, kSlabMin);
        char* base = nullptr;
        hipError_t e = hipMalloc((void**)&base, sz);
        if (e != hipSuccess && sz > bytes) { sz = bytes; e = hipMalloc((void**)&base, sz); }
        if (e != hipSuccess)
            throw Error(SDMI_ERR_HIP, std::string("DevPool: hipMalloc(") + std::to_string(sz) + ") failed: " + hipGetErrorString(e));
        Slab s; s.base = base; s.size = sz; s.free_list.push_back({0, sz});
        slabs_.push_back(std::move(s));
        reserved_ += sz;
    }
    throw Error(SDMI_ERR_HIP, "DevPool: allocation failed");
}

void DevPool::free(void* p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) throw Error(SDMI_ERR_STATE, "DevPool: free of unknown pointer");
    const int si = it->second.slab;
    const size_t size = it->second.size;
    live_.erase(it);
    in_use_ -= size;
    auto& sl = slabs_[si];
    const size_t off = (char*)p - sl.base;
    auto& fl = sl.free_list;
    size_t pos = 0;
    while (pos < fl.size() && fl[pos].off < off) ++pos;
    fl.insert(fl.begin() + pos, {off, size});
    if (pos + 1 < fl.size() && fl[pos].off + fl[pos].size == fl[pos + 1].off) {
        fl[pos].size += fl[pos + 1].size;
        fl.erase(fl.begin() + pos + 1);
    }
    if (pos > 0 && fl[pos - 1].off + fl[pos - 1].size == fl[pos].off) {
        fl[pos - 1].size += fl[pos].size;
        fl.erase(fl.begin() + pos);
    }
}

size_t DevPool::free_since(unsigned long long mark) {
    std::vector<void*> victims;
    for (auto& kv : live_)
        if (kv.second.serial > mark) victims.push_back(kv.first);
    for (void* p : victims) free(p);
    return victims.size();
}

// =============================================================================
// per-class kernel timing (HIP events on the engine stream)
// =============================================================================
hipEvent_t Engine::prof_event() {
    if (!prof_free_.empty()) { hipEvent_t e = prof_free_.back(); prof_free_.pop_back(); return e; }
    hipEvent_t e;
    SDMI_HIP(hipEventCreate(&e));
    return e;
}

Engine::ProfScope::ProfScope(Engine* e_, int cls_, double flops_, double bytes_, int n_launch_) : e(e_), cls(cls_), flops(flops_), bytes(bytes_), n_launch(n_launch_) {
    if (!e->profiling_) return;
    a = e->prof_event();
    b = e->prof_event();
    (void)hipEventRecord(a, e->stream_);
}

Engine::ProfScope::~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, e->stream_);
    e->prof_pending_.push_back({cls, a, b, flops, bytes, n_launch, tag});
    if (e->prof_pending_.size() >= 2048) {
        try { e->prof_flush(); } catch (...) {}
    }
}

void Engine::ProfScope::set_tag(const char* fmt, ...) {
    if (!a || !e->prof_tagging_) return;
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    auto it = e->prof_tag_ids_.find(buf);
    if (it == e->prof_tag_ids_.end()) {
        it = e->prof_tag_ids_.emplace(buf, (int)e->prof_tag_names_.size()).first;
        e->prof_tag_names_.push_back(buf);
    }
    tag = it->second;
}

// What an (a, b) event pair reads with NOTHING between its two records: the pair's own cost on the stream, which would otherwise be
// booked as kernel time on every launch (round 2: the classes summed to 7 % more than the step).  Median of 64 empty pairs, taken with the
// stream otherwise idle when profiling is switched on, and subtracted from every sample.
void Engine::prof_calibrate() {
    SDMI_HIP(hipStreamSynchronize(stream_));
    std::vector<float> t;
    for (int i = 0; i < 64; ++i) {
        hipEvent_t a = prof_event(), b = prof_event();
        SDMI_HIP(hipEventRecord(a, stream_));
        SDMI_HIP(hipEventRecord(b, stream_));
        SDMI_HIP(hipEventSynchronize(b));
        float ms = 0;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess) t.push_back(ms);
        prof_free_.push_back(a);
        prof_free_.push_back(b);
    }
    std::sort(t.begin(), t.end());
    // An empty pair reads the cost of TWO back-to-back event records (4.6 us on MI355X); a pair around a kernel adds about half of that
    // to the kernel's duration -- rocprofv3's kernel trace of the same run is the yardstick: round 2, 46.0 us by raw events against 43.4 us
    // by rocprofv3 per split-GEMM launch; round 3, 43.4 against 41.1 (profiles/README.md).  So half the empty-pair reading is subtracted.
    prof_overhead_ms_ = t.empty() ? 0.0 : 0.5 * (double)t[t.size() / 2];
}

void Engine::prof_flush() {
    if (prof_pending_.empty()) return;
    SDMI_HIP(hipStreamSynchronize(stream_));
    for (auto& p : prof_pending_) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ms = (float)std::max(0.0, (double)ms - prof_overhead_ms_);
            prof_[p.cls].ms += ms;
            prof_[p.cls].launches += p.n_launch;
            prof_[p.cls].flops += p.flops;
            prof_[p.cls].bytes += p.bytes;
            if (p.tag >= 0) {
                ProfStat& t = prof_tags_[prof_tag_names_[p.tag]];
                t.ms += ms; t.launches += p.n_launch; t.flops += p.flops; t.bytes += p.bytes;
            }
        }
        prof_free_.push_back(p.a);
        prof_free_.push_back(p.b);
    }
    prof_pending_.clear();
}

// =============================================================================
// roctx ranges
// =============================================================================
namespace {
struct Roctx {
    void* lib = nullptr;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool tried = false;
} g_roctx;
}  // namespace

void Engine::roctx_enable(bool on) {
    if (on && !g_roctx.tried) {
        g_roctx.tried = true;
        for (const char* name : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/libroctx64.so"}) {
            g_roctx.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (g_roctx.lib) break;
        }
        if (g_roctx.lib) {
            g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(g_roctx.lib, "roctxRangePushA"));
            g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(g_roctx.lib, "roctxRangePop"));
        }
    }
    if (on && (!g_roctx.push || !g_roctx.pop)) throw Error(SDMI_ERR_UNSUPPORTED, "roctx: libroctx64 (roctxRangePushA / roctxRangePop) not found");
    roctx_on_ = on;
}
void Engine::roctx_push(const char* name) { if (g_roctx.push) (void)g_roctx.push(name); }
void Engine::roctx_pop() { if (g_roctx.pop) (void)g_roctx.pop(); }

// =============================================================================
// construction / model definition
// =============================================================================
Engine::Engine(const sdmi_config& cfg) : cfg_(cfg) {
    if (cfg.precision < 0 || cfg.precision > 2) throw Error(SDMI_ERR_UNSUPPORTED, "precision must be 0 (fp32), 1 (bf16) or 2 (bf16 + MXFP8 ResBlock convolutions)");
    bf16_ = cfg.precision >= 1;
    fp8_ = cfg.precision == 2;
    if (bf16_ && (cfg.model_channels % 64 || cfg.vae_ch % 64 || cfg.ctx_dim % 64))
        throw Error(SDMI_ERR_UNSUPPORTED, "precision=1 (bf16) needs model_channels, vae_ch and ctx_dim to be multiples of 64");
    if (cfg.model_channels % 32 || cfg.model_channels <= 0) throw Error(SDMI_ERR_INVALID, "model_channels must be a positive multiple of 32");
    if (cfg.vae_ch % 32 || cfg.vae_ch <= 0) throw Error(SDMI_ERR_INVALID, "vae_ch must be a positive multiple of 32");
    if (cfg.n_head <= 0 || cfg.model_channels % cfg.n_head) throw Error(SDMI_ERR_INVALID, "n_head must divide model_channels");
    if (cfg.ctx_dim % 32 || cfg.ctx_dim <= 0) throw Error(SDMI_ERR_INVALID, "ctx_dim must be a positive multiple of 32");
    if (cfg.latent_h % 8 || cfg.latent_w % 8 || cfg.latent_h <= 0 || cfg.latent_w <= 0)
        throw Error(SDMI_ERR_INVALID, "latent_h/latent_w must be positive multiples of 8");
    int ndev = 0;
    SDMI_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) throw Error(SDMI_ERR_HIP, "no HIP device visible: libsdmi has no CPU fallback");
    if (cfg.device < 0 || cfg.device >= ndev) throw Error(SDMI_ERR_INVALID, "device ordinal out of range");
    SDMI_HIP(hipSetDevice(cfg.device));
    hipDeviceProp_t prop;
    SDMI_HIP(hipGetDeviceProperties(&prop, cfg.device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        throw Error(SDMI_ERR_UNSUPPORTED, std::string("libsdmi is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    try {
    SDMI_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    SDMI_HIP(hipEventCreate(&ev0_));
    SDMI_HIP(hipEventCreate(&ev1_));
    SDMI_HIP(hipEventCreateWithFlags(&ev_user_, hipEventDisableTiming));
    SDMI_HIP(hipMalloc(&zero_page_, 256));
    weight_allocs_.push_back(zero_page_);
    SDMI_HIP(hipMemsetAsync(zero_page_, 0, 256, stream_));
    SDMI_HIP(hipStreamSynchronize(stream_));
    {   // measured per-shape tile choices (tools/autotune.py -> tuning/gfx950_fp32.txt)
        struct Row { const char* key; int cfg; int splits; };
        static const Row rows[] = {
#include "tuning_table.inc"
            {nullptr, 0, 0}};
        for (const Row* r = rows; r->key; ++r) tuned_[r->key] = TileChoice{r->cfg, r->splits};
        static const Row rows_mfma[] = {
#include "tuning_table_mfma.inc"
            {nullptr, 0, 0}};
        for (const Row* r = rows_mfma; r->key; ++r) tuned_mfma_[r->key] = TileChoice{r->cfg, r->splits};
        static const Row rows_p[] = {
#include "tuning_table_planes.inc"
            {nullptr, 0, 0}};
        for (const Row* r = rows_p; r->key; ++r) tuned_p_[r->key] = TileChoice{r->cfg, r->splits};
        static const Row rows16[] = {
#include "tuning_table_bf16.inc"
            {nullptr, 0, 0}};
        for (const Row* r = rows16; r->key; ++r) tuned_bf16_[r->key] = TileChoice{r->cfg, r->splits};
    }
    build_model();
    } catch (...) {   // the destructor does not run for a half-built object
        destroy();
        throw;
    }
}

Engine::~Engine() { destroy(); }

void Engine::destroy() noexcept {
    (void)hipSetDevice(cfg_.device);
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (void* p : weight_allocs_) (void)hipFree(p);
    for (auto& p : prof_pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : prof_free_) (void)hipEventDestroy(e);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (ev_user_) (void)hipEventDestroy(ev_user_);
    if (stream_) (void)hipStreamDestroy(stream_);
    weight_allocs_.clear(); prof_pending_.clear(); prof_free_.clear();
    ev0_ = ev1_ = ev_user_ = nullptr; stream_ = nullptr;
}

void Engine::add_entry(const std::string& name, int kind, std::initializer_list<int64_t> dims, float** dst, int wdt) {
    WeightEntry e;
    e.name = name; e.kind = kind; e.ndim = (int)dims.size(); e.dst = dst; e.wdt = wdt; e.group = cur_group_;
    int i = 0;
    for (int k = 0; k < 4; ++k) e.dims[k] = 1;
    for (auto d : dims) e.dims[i++] = d;
    entry_index_[name] = (int)entries_.size();
    entries_.push_back(e);
}

void Engine::add_meta(const std::string& name, int n, float e0, float e1, float* store) {
    MetaEntry m{name, n, {e0, e1}, store};
    meta_index_[name] = (int)meta_.size();
    meta_.push_back(m);
}

// The structs build_model() fills live in members / vectors whose storage is fixed
// before any entry is added (entries_ keeps float** into them).
void Engine::build_model() {
    const int mc = cfg_.model_channels, ed = 4 * mc, cd = cfg_.ctx_dim;
    const int c1 = mc, c2 = 2 * mc, c4 = 4 * mc;
    auto add = [](Engine* e, const std::string& n, int kind, std::initializer_list<int64_t> d, float** dst, int wdt = 0) { e->add_entry(n, kind, d, dst, wdt); };
    // precision = 1: weights of every layer with Cin % 64 == 0 are packed as bf16; the three Cin = 4
    // layers and the time-embedding MLPs (M = n_steps rows, once per call) stay fp32
    // stride / pad: what the engine applies to this layer; checked against the dump's metadata files (load_conv2d, load.rs:118-160)
    auto conv = [&](ConvW& w, const std::string& path, int cin, int cout, int k, int stride = 1, int pad = -1) {
        w.cin = cin; w.cout = cout; w.k = k;
        w.dt = (bf16_ && cin % 64 == 0) ? 1 : 0;
        if (bf16_ && !w.dt && cin >= 32) throw Error(SDMI_ERR_UNSUPPORTED, "bf16: conv Cin must be a multiple of 64 (or < 32)");
        add(this, path + "/weight", 0, {cout, cin, k, k}, &w.bt, w.dt);
        add(this, path + "/bias", 2, {cout}, &w.bias);
        if (pad < 0) pad = k == 3 ? 1 : 0;
        add_meta(path + "/stride", 2, (float)stride, (float)stride, nullptr);
        add_meta(path + "/padding", 2, (float)pad, (float)pad, nullptr);
        add_meta(path + "/dilation", 2, 1.f, 1.f, nullptr);
        add_meta(path + "/kernel_size", 2, (float)k, (float)k, nullptr);
        add_meta(path + "/n_group", 1, 1.f, 0.f, nullptr);
    };
    auto lin = [&](LinW& w, const std::string& path, int cin, int cout, bool bias, bool keep_f32 = false) {
        w.cin = cin; w.cout = cout;
        w.dt = (bf16_ && !keep_f32) ? 1 : 0;
        if (w.dt && cin % 64) throw Error(SDMI_ERR_UNSUPPORTED, "bf16: linear in_features must be a multiple of 64");
        add(this, path + "/weight", 1, {cin, cout}, &w.bt, w.dt);
        if (bias) add(this, path + "/bias", 2, {cout}, &w.bias);
    };
    auto norm = [&](NormW& w, const std::string& path, int c, bool group_norm = true) {
        w.c = c;
        add(this, path + "/weight", 2, {c}, &w.gamma);
        add(this, path + "/bias", 2, {c}, &w.beta);
        add_meta(path + "/eps", 1, 0.f, 0.f, &w.eps);
        if (group_norm) {
            add_meta(path + "/n_group", 1, 32.f, 0.f, nullptr);   // GroupNormConfig::new(32, ..) everywhere (unet/mod.rs:692, autoencoder/mod.rs:483)
            add_meta(path + "/n_channel", 1, (float)c, 0.f, nullptr);
        }
    };
    // precision = 2: these convs (fed by a GroupNorm + SiLU, Cin % 32 == 0, Cout % 8 == 0) also get an MXFP8 copy of their weight
    auto fp8_copy = [&](ConvW& w, const std::string& weight_name) {
        if (!fp8_ || w.cin % 32 || w.cout % 8 || !w.dt) return;
        WeightEntry& e = entries_[entry_index_.at(weight_name)];
        e.dst8 = &w.bt8;
        e.dsts = &w.bs8;
    };
    // ... and (option fp8_linear) the Linear layers of the transformer blocks
    auto fp8_copy_lin = [&](LinW& w, const std::string& weight_name) {
        if (!fp8_ || w.cin % 32 || w.cout % 8 || !w.dt) return;
        WeightEntry& e = entries_[entry_index_.at(weight_name)];
        e.dst8 = &w.bt8;
        e.dsts = &w.bs8;
    };
    auto res = [&](ResW& r, const std::string& path, int cin, int cout, bool unet) {
        r.cin = cin; r.cout = cout; r.has_embed = unet; r.has_skip = cin != cout;
        if (unet) {  // ResBlock, unet/mod.rs:679-734; names unet/load.rs:20-25
            norm(r.norm_in, path + "/norm_in", cin);
            conv(r.conv_in, path + "/conv_in", cin, cout, 3);
            lin(r.lin_embed, path + "/lin_embed", ed, cout, true, /*keep_f32=*/true);
            norm(r.norm_out, path + "/norm_out", cout);
            conv(r.conv_out, path + "/conv_out", cout, cout, 3);
            if (r.has_skip) conv(r.skip, path + "/skip_connection", cin, cout, 1);
            fp8_copy(r.conv_in, path + "/conv_in/weight");
            fp8_copy(r.conv_out, path + "/conv_out/weight");
            if (r.has_skip) fp8_copy(r.skip, path + "/skip_connection/weight");
        } else {  // ResnetBlock, autoencoder/mod.rs:472-528; names autoencoder/load.rs:39-45
            norm(r.norm_in, path + "/norm1", cin);
            conv(r.conv_in, path + "/conv1", cin, cout, 3);
            norm(r.norm_out, path + "/norm2", cout);
            conv(r.conv_out, path + "/conv2", cout, cout, 3);
            if (r.has_skip) conv(r.skip, path + "/nin_shortcut", cin, cout, 1);
            fp8_copy(r.conv_in, path + "/conv1/weight");
            fp8_copy(r.conv_out, path + "/conv2/weight");
            // (the decoder's 1x1 shortcuts and up-convolutions stay bf16 under fp8_linear too: measured on MI355X, MXFP8 there moves the decoded
            // RGB from 2.1e-2 to 9.8e-2 relative RMS of the exact decode and saves 0.4 ms per image)
        }
    };
    auto mha = [&](MhaW& m, const std::string& path, int c, int cctx) {  // unet/mod.rs:603-653
        if (c == cctx) {
            // self-attention: query/key/value weights are packed into ONE [3c][c] buffer so the three
            // projections of unet/mod.rs:645-647 run as a single GEMM with N = 3c
            void* p = nullptr;
            SDMI_HIP(hipMalloc(&p, (size_t)3 * c * c * esz()));
            weight_allocs_.push_back(p);
            m.q.bt = reinterpret_cast<float*>(p);
            if (!bf16_) {   // and its bf16 planes (k_gemm3x.hip), same 1.5x offset rule as the arenas
                void* pl = nullptr;
                SDMI_HIP(hipMalloc(&pl, (size_t)3 * c * c * 6));
                weight_allocs_.push_back(pl);
                split_regions_.push_back(SplitRegion{reinterpret_cast<char*>(p), (size_t)3 * c * c * 4, reinterpret_cast<char*>(pl)});
            }
            m.k.bt = adv(m.q.bt, (long long)c * c, edt());
            m.v.bt = adv(m.q.bt, (long long)2 * c * c, edt());
            if (fp8_ && c % 32 == 0) {   // the packed [3c][Kp] MXFP8 copy + its scales: one N = 3c GEMM as well
                const size_t kp = (size_t)(c + 127) / 128 * 128;
                void *q8 = nullptr, *s8 = nullptr;
                SDMI_HIP(hipMalloc(&q8, (size_t)3 * c * kp));
                weight_allocs_.push_back(q8);
                SDMI_HIP(hipMalloc(&s8, (size_t)3 * c * kp / 32));
                weight_allocs_.push_back(s8);
                m.q.bt8 = reinterpret_cast<float*>(q8);
                m.k.bt8 = reinterpret_cast<float*>((char*)q8 + (size_t)c * kp);
                m.v.bt8 = reinterpret_cast<float*>((char*)q8 + (size_t)2 * c * kp);
                m.q.bs8 = reinterpret_cast<float*>(s8);
                m.k.bs8 = reinterpret_cast<float*>((char*)s8 + (size_t)c * kp / 32);
                m.v.bs8 = reinterpret_cast<float*>((char*)s8 + (size_t)2 * c * kp / 32);
            }
        }
        lin(m.q, path + "/query", c, c, false);
        // precision >= 1: the bf16 attention kernel takes q in log2 units -- d_head^-0.5 log2(e) is folded into the query weight here, in
        // fp32, before its only rounding (attention.rs:15-26 applies d_head^-0.25 to q and to k)
        if (q_prescaled(m.q.dt, c / cfg_.n_head)) entries_[entry_index_.at(path + "/query/weight")].pre_scale = attn_bf16_q_scale(c / cfg_.n_head);
        lin(m.k, path + "/key", cctx, c, false);
        lin(m.v, path + "/value", cctx, c, false);
        lin(m.out, path + "/out", c, c, true);
        fp8_copy_lin(m.q, path + "/query/weight");
        if (c == cctx) { fp8_copy_lin(m.k, path + "/key/weight"); fp8_copy_lin(m.v, path + "/value/weight"); }   // (cross-attention K / V: hoisted, once per call, bf16)
        fp8_copy_lin(m.out, path + "/out/weight");
        add_meta(path + "/n_head", 1, (float)cfg_.n_head, 0.f, nullptr);
    };
    auto spatial = [&](SpatialW& s, const std::string& path, int c) {  // unet/mod.rs:436-527
        s.c = c;
        norm(s.norm, path + "/norm", c);
        conv(s.proj_in, path + "/proj_in", c, c, 1);
        const std::string t = path + "/transformer";
        norm(s.ln1, t + "/norm1", c, false);
        mha(s.attn1, t + "/attn1", c, c);
        norm(s.ln2, t + "/norm2", c, false);
        mha(s.attn2, t + "/attn2", c, cd);
        norm(s.ln3, t + "/norm3", c, false);
        lin(s.geglu_proj, t + "/mlp/geglu/proj", c, 8 * c, true);
        lin(s.mlp_lin, t + "/mlp/lin", 4 * c, c, true);
        conv(s.proj_out, path + "/proj_out", c, c, 1);
        fp8_copy(s.proj_in, path + "/proj_in/weight");
        fp8_copy(s.proj_out, path + "/proj_out/weight");
        fp8_copy_lin(s.geglu_proj, t + "/mlp/geglu/proj/weight");
        fp8_copy_lin(s.mlp_lin, t + "/mlp/lin/weight");
    };

    add(this, "alphas_cumprod", 3, {1000}, nullptr);
    add_meta("n_steps", 1, 1000.f, 0.f, nullptr);   // stablediffusion/load.rs:20

    // ---- UNet (unet/mod.rs:36-92) ------------------------------------------------
    lin(lin1_time_, "unet/lin1_time_embed", mc, ed, true, /*keep_f32=*/true);
    lin(lin2_time_, "unet/lin2_time_embed", ed, ed, true, /*keep_f32=*/true);
    struct Spec { BlockKind kind; const char* name; int cin, cout; };
    const Spec in_spec[12] = {
        {BK_CONV, "conv", 4, c1},   {BK_RES_ST, "rt1", c1, c1}, {BK_RES_ST, "rt2", c1, c1}, {BK_DOWN, "d1", c1, c1},
        {BK_RES_ST, "rt3", c1, c2}, {BK_RES_ST, "rt4", c2, c2}, {BK_DOWN, "d2", c2, c2},    {BK_RES_ST, "rt5", c2, c4},
        {BK_RES_ST, "rt6", c4, c4}, {BK_DOWN, "d3", c4, c4},    {BK_RES, "r1", c4, c4},     {BK_RES, "r2", c4, c4}};
    const Spec out_spec[12] = {
        {BK_RES, "r1", 2 * c4, c4},        {BK_RES, "r2", 2 * c4, c4},        {BK_RES_UP, "ru", 2 * c4, c4},
        {BK_RES_ST, "rt1", 2 * c4, c4},    {BK_RES_ST, "rt2", 2 * c4, c4},    {BK_RES_ST_UP, "rtu1", c4 + c2, c4},
        {BK_RES_ST, "rt3", c4 + c2, c2},   {BK_RES_ST, "rt4", 2 * c2, c2},    {BK_RES_ST_UP, "rtu2", c2 + c1, c2},
        {BK_RES_ST, "rt5", c2 + c1, c1},   {BK_RES_ST, "rt6", 2 * c1, c1},    {BK_RES_ST, "rt7", 2 * c1, c1}};
    in_blocks_.resize(12);
    out_blocks_.resize(12);
    auto def_block = [&](UBlock& b, const Spec& s, const std::string& root) {
        b.kind = s.kind; b.cin = s.cin; b.cout = s.cout;
        const std::string path = root + "/" + s.name;
        switch (s.kind) {
            case BK_CONV: conv(b.conv, path, s.cin, s.cout, 3); break;
            case BK_DOWN: conv(b.conv, path, s.cin, s.cout, 3, 2); fp8_copy(b.conv, path + "/weight"); break;  // load_downsample: path itself (unet/load.rs:138-143); stride 2, pad 1 (unet/mod.rs:413-418)
            case BK_RES: res(b.res, path, s.cin, s.cout, true); break;
            default:
                res(b.res, path + "/res", s.cin, s.cout, true);
                if (s.kind == BK_RES_ST || s.kind == BK_RES_ST_UP) spatial(b.st, path + "/transformer", s.cout);
                if (s.kind == BK_RES_UP || s.kind == BK_RES_ST_UP) { conv(b.up, path + "/upsample/conv", s.cout, s.cout, 3); fp8_copy(b.up, path + "/upsample/conv/weight"); }
        }
    };
    for (int i = 0; i < 12; ++i) def_block(in_blocks_[i], in_spec[i], "unet/input_blocks");
    res(mid_res1_, "unet/middle_block/res1", c4, c4, true);
    spatial(mid_st_, "unet/middle_block/transformer", c4);
    res(mid_res2_, "unet/middle_block/res2", c4, c4, true);
    for (int i = 0; i < 12; ++i) def_block(out_blocks_[i], out_spec[i], "unet/output_blocks");
    norm(unet_norm_out_, "unet/norm_out", c1);
    conv(unet_conv_out_, "unet/conv_out", c1, 4, 3);

    // index ResBlocks / SpatialTransformers for the hoisted per-call tables
    auto idx_res = [&](ResW& r) { r.temb_index = (int)res_list_.size(); res_list_.push_back(&r); };
    auto idx_st = [&](SpatialW& s) { s.ctx_index = (int)st_list_.size(); st_list_.push_back(&s); };
    auto idx_block = [&](UBlock& b) {
        if (b.kind == BK_CONV || b.kind == BK_DOWN) return;
        idx_res(b.res);
        if (b.kind == BK_RES_ST || b.kind == BK_RES_ST_UP) idx_st(b.st);
    };
    for (auto& b : in_blocks_) idx_block(b);
    idx_res(mid_res1_); idx_st(mid_st_); idx_res(mid_res2_);
    for (auto& b : out_blocks_) idx_block(b);

    // ---- VAE decoder (autoencoder/mod.rs:30-36,154-191) ------------------------------
    const int vc = cfg_.vae_ch;
    const int dch[4][2] = {{4 * vc, 4 * vc}, {4 * vc, 4 * vc}, {4 * vc, 2 * vc}, {2 * vc, vc}};
    conv(post_quant_, "autoencoder/post_quant_conv", 4, 4, 1);
    conv(dec_conv_in_, "autoencoder/decoder/conv_in", 4, 4 * vc, 3);
    res(dec_mid1_, "autoencoder/decoder/mid/block_1", 4 * vc, 4 * vc, false);
    dec_attn_.c = 4 * vc;
    norm(dec_attn_.norm, "autoencoder/decoder/mid/attn/norm", 4 * vc);
    conv(dec_attn_.q, "autoencoder/decoder/mid/attn/q", 4 * vc, 4 * vc, 1);
    conv(dec_attn_.k, "autoencoder/decoder/mid/attn/k", 4 * vc, 4 * vc, 1);
    conv(dec_attn_.v, "autoencoder/decoder/mid/attn/v", 4 * vc, 4 * vc, 1);
    conv(dec_attn_.proj_out, "autoencoder/decoder/mid/attn/proj_out", 4 * vc, 4 * vc, 1);
    res(dec_mid2_, "autoencoder/decoder/mid/block_2", 4 * vc, 4 * vc, false);
    for (int i = 0; i < 4; ++i) {
        DecBlockW& b = dec_blocks_[i];
        b.cin = dch[i][0]; b.cout = dch[i][1]; b.has_up = i != 3;
        const std::string bp = "autoencoder/decoder/blocks/" + std::to_string(i);
        res(b.res[0], bp + "/res1", b.cin, b.cout, false);
        res(b.res[1], bp + "/res2", b.cout, b.cout, false);
        res(b.res[2], bp + "/res3", b.cout, b.cout, false);
        if (b.has_up) conv(b.upsampler, bp + "/upsampler", b.cout, b.cout, 3);
    }
    norm(dec_norm_out_, "autoencoder/decoder/norm_out", vc);
    conv(dec_conv_out_, "autoencoder/decoder/conv_out", vc, 3, 3);

    // ---- CLIP text encoder (clip/mod.rs:18-45; dump names clip/load.rs:14-91) -- SURVEY 8f rank 2 -------------
    // An optional weight group: the sampling path takes the text embedding as an input, so a context without
    // CLIP weights is complete; sdmi_clip_forward / sdmi_context need the whole group.  Always fp32.
    if (cfg_.clip_layers > 0) {
        const int cs = cd, L = cfg_.clip_layers, H = cfg_.clip_heads;
        if (H <= 0 || cs % H || !attn_supported_head_dim(cs / H) || cfg_.clip_vocab <= 0 || cfg_.clip_ctx <= 0 || cs % 32)
            throw Error(SDMI_ERR_UNSUPPORTED, "CLIP: ctx_dim / clip_heads must be one of the fused attention head dims (40, 64, 80, 160)");
        cur_group_ = 1;
        add(this, "clip/token_embedding/weight", 2, {cfg_.clip_vocab, cs}, &clip_tok_);
        add(this, "clip/position_embedding/weight", 2, {cfg_.clip_ctx, cs}, &clip_pos_);
        clip_blocks_.resize(L);
        for (int i = 0; i < L; ++i) {
            ClipBlockW& b = clip_blocks_[i];
            const std::string bp = "clip/blocks/" + std::to_string(i);
            void* w = nullptr;
            void* bias = nullptr;
            SDMI_HIP(hipMalloc(&w, (size_t)3 * cs * cs * sizeof(float)));
            weight_allocs_.push_back(w);
            SDMI_HIP(hipMalloc(&bias, (size_t)3 * cs * sizeof(float)));
            weight_allocs_.push_back(bias);
            b.q.bt = reinterpret_cast<float*>(w); b.k.bt = b.q.bt + (size_t)cs * cs; b.v.bt = b.q.bt + (size_t)2 * cs * cs;
            b.q.bias = reinterpret_cast<float*>(bias); b.k.bias = b.q.bias + cs; b.v.bias = b.q.bias + 2 * cs;
            norm(b.attn_ln, bp + "/attn_ln", cs, false);
            lin(b.q, bp + "/attn/query", cs, cs, true, true);   // MultiHeadSelfAttention: all four Linears carry a bias
            lin(b.k, bp + "/attn/key", cs, cs, true, true);
            lin(b.v, bp + "/attn/value", cs, cs, true, true);
            lin(b.out, bp + "/attn/out", cs, cs, true, true);
            norm(b.mlp_ln, bp + "/mlp_ln", cs, false);
            lin(b.fc1, bp + "/mlp/fc1", cs, 4 * cs, true, true);
            lin(b.fc2, bp + "/mlp/fc2", 4 * cs, cs, true, true);
        }
        norm(clip_ln_, "clip/layer_norm", cs, false);
        cur_group_ = 0;
    }

    // ---- VAE encoder (autoencoder/mod.rs:30-32,76-144,220-265; names autoencoder/load.rs:120-190) -- SURVEY 8f rank 4
    // Optional weight group: `sample` never encodes; sdmi_encode_image needs the whole group.
    {
        cur_group_ = 2;
        const int ech[4][2] = {{vc, vc}, {vc, 2 * vc}, {2 * vc, 4 * vc}, {4 * vc, 4 * vc}};
        enc_conv_in_.cin = 4; enc_conv_in_.cout = vc; enc_conv_in_.k = 3; enc_conv_in_.dt = 0;   // RGB + one zero channel
        add(this, "autoencoder/encoder/conv_in/weight", 0, {vc, 3, 3, 3}, &enc_conv_in_.bt);
        add(this, "autoencoder/encoder/conv_in/bias", 2, {vc}, &enc_conv_in_.bias);
        for (int i = 0; i < 4; ++i) {
            EncBlockW& b = enc_blocks_[i];
            b.cin = ech[i][0]; b.cout = ech[i][1]; b.has_down = i != 3;
            const std::string bp = "autoencoder/encoder/blocks/" + std::to_string(i);
            res(b.res[0], bp + "/res1", b.cin, b.cout, false);
            res(b.res[1], bp + "/res2", b.cout, b.cout, false);
            if (b.has_down) conv(b.down, bp + "/downsampler/conv", b.cout, b.cout, 3, 2, 0);   // load_padded_conv2d: "{path}/conv", saved with padding (0, 0) (python/save.py:73-76)
        }
        res(enc_mid1_, "autoencoder/encoder/mid/block_1", 4 * vc, 4 * vc, false);
        enc_attn_.c = 4 * vc;
        norm(enc_attn_.norm, "autoencoder/encoder/mid/attn/norm", 4 * vc);
        conv(enc_attn_.q, "autoencoder/encoder/mid/attn/q", 4 * vc, 4 * vc, 1);
        conv(enc_attn_.k, "autoencoder/encoder/mid/attn/k", 4 * vc, 4 * vc, 1);
        conv(enc_attn_.v, "autoencoder/encoder/mid/attn/v", 4 * vc, 4 * vc, 1);
        conv(enc_attn_.proj_out, "autoencoder/encoder/mid/attn/proj_out", 4 * vc, 4 * vc, 1);
        res(enc_mid2_, "autoencoder/encoder/mid/block_2", 4 * vc, 4 * vc, false);
        norm(enc_norm_out_, "autoencoder/encoder/norm_out", 4 * vc);
        conv(enc_conv_out_, "autoencoder/encoder/conv_out", 4 * vc, 8, 3);
        conv(quant_conv_, "autoencoder/quant_conv", 8, 8, 1);
        cur_group_ = 0;
    }
}

// =============================================================================
// weights
// =============================================================================
// Loading is batched: every tensor goes host -> pinned ring -> (device staging ->) packing kernel -> its slot of
// ONE device arena per weight group, all asynchronous on the engine's stream; the callers synchronise once
// (load_weights_dir / load_weights_packed) or per tensor (set_weight, whose source may be freed on return).
// The reference builds ~1100 Burn tensors one file at a time (stablediffusion/load.rs:16-33, model/load.rs:30-46).
static constexpr size_t kStageBytes = (size_t)160 << 20;   // >= the largest tensor (CLIP token table, 152 MB)

struct Engine::Stager {
    char* pinned[2] = {nullptr, nullptr};
    char* dev[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    size_t used = 0;
    int cur = 0;
    ~Stager() {
        for (int i = 0; i < 2; ++i) {
            if (done[i]) { (void)hipEventSynchronize(done[i]); (void)hipEventDestroy(done[i]); }
            if (pinned[i]) (void)hipHostFree(pinned[i]);
            if (dev[i]) (void)hipFree(dev[i]);
        }
    }
};

void Engine::stager_release() { stager_.reset(); }

// `bytes` of pinned host memory (256-byte aligned) the caller fills; valid until the matching commit
char* Engine::stage_reserve(size_t bytes, size_t* offset, int* half) {
    if (bytes > kStageBytes) throw Error(SDMI_ERR_UNSUPPORTED, "weight tensor larger than the staging buffer");
    if (!stager_) {
        stager_.reset(new Stager());
        for (int i = 0; i < 2; ++i) {
            SDMI_HIP(hipHostMalloc((void**)&stager_->pinned[i], kStageBytes, hipHostMallocDefault));
            SDMI_HIP(hipMalloc((void**)&stager_->dev[i], kStageBytes));
            SDMI_HIP(hipEventCreateWithFlags(&stager_->done[i], hipEventDisableTiming));
        }
    }
    Stager& st = *stager_;
    const size_t need = (bytes + 255) / 256 * 256;
    if (st.used + need > kStageBytes) {   // this half is full: mark it in flight, move to the other one
        SDMI_HIP(hipEventRecord(st.done[st.cur], stream_));
        st.busy[st.cur] = true;
        st.cur ^= 1;
        st.used = 0;
        if (st.busy[st.cur]) { SDMI_HIP(hipEventSynchronize(st.done[st.cur])); st.busy[st.cur] = false; }
    }
    *offset = st.used;
    *half = st.cur;
    st.used += need;
    return st.pinned[st.cur] + *offset;
}

void Engine::ensure_arena(int group) {
    if (arena_done_[group]) return;
    size_t total = 0;
    for (auto& e : entries_) {
        if (e.group != group || e.kind == 3 || *e.dst) continue;
        size_t count = 1;
        for (int i = 0; i < e.ndim; ++i) count *= (size_t)e.dims[i];
        if (e.kind == 0 && e.dims[1] == 3) count = count / 3 * 4;   // RGB conv_in: packed with a zero 4th input channel
        total += (count * (e.wdt ? 2 : 4) + 255) / 256 * 256;
    }
    if (total) {
        char* base = nullptr;
        SDMI_HIP(hipMalloc((void**)&base, total));
        weight_allocs_.push_back(base);
        arena_base_[group] = base;
        arena_bytes_[group] = total;
        if (!bf16_) {
            SDMI_HIP(hipMalloc((void**)&split_base_[group], total / 2 * 3));
            weight_allocs_.push_back(split_base_[group]);
        }
        size_t off = 0;
        for (auto& e : entries_) {
            if (e.group != group || e.kind == 3 || *e.dst) continue;
            size_t count = 1;
            for (int i = 0; i < e.ndim; ++i) count *= (size_t)e.dims[i];
            if (e.kind == 0 && e.dims[1] == 3) count = count / 3 * 4;
            *e.dst = reinterpret_cast<float*>(base + off);
            off += (count * (e.wdt ? 2 : 4) + 255) / 256 * 256;
        }
    }
    arena_done_[group] = true;
}

Engine::TempSplit::TempSplit(Engine* e_, const float* bt_, long long rows, long long K) : e(e_), bt(bt_) {
    if (e->bf16_ || K % 32 || rows <= 0) return;
    planes = e->pool_.alloc((size_t)rows * K * 6);
    hipError_t err = launch_pack_split3(bt, planes, rows, (int)K, e->stream_, e->b3_grouped(rows));
    if (err != hipSuccess) { e->pool_.free(planes); planes = nullptr; SDMI_HIP(err); }
    e->temp_split_bt_ = bt;
    e->temp_split_planes_ = planes;
}
Engine::TempSplit::~TempSplit() {
    if (!planes) return;
    e->temp_split_bt_ = nullptr;
    e->temp_split_planes_ = nullptr;
    e->pool_.free(planes);
}

const void* Engine::split_planes(const float* bt) const {
    if (bt && bt == temp_split_bt_) return temp_split_planes_;
    const char* b = reinterpret_cast<const char*>(bt);
    for (int g = 0; g < 3; ++g)
        if (split_base_[g] && b >= arena_base_[g] && b < arena_base_[g] + arena_bytes_[g]) return split_base_[g] + (size_t)(b - arena_base_[g]) / 2 * 3;
    for (const SplitRegion& r : split_regions_)    // the packed q | k | v weights of the self-attentions (own allocations)
        if (b >= r.base && b < r.base + r.bytes) return r.planes + (size_t)(b - r.base) / 2 * 3;
    return nullptr;
}

static size_t entry_count(const WeightEntry& e) {
    size_t count = 1;
    for (int i = 0; i < e.ndim; ++i) count *= (size_t)e.dims[i];
    return count;
}

// Enqueues the upload + packing of one tensor whose fp32 values (reference layout) the caller has written to the
// pinned block (half, offset) returned by stage_reserve.
void Engine::stage_commit(WeightEntry& e, size_t offset, int half) {
    Stager& st = *stager_;
    const size_t count = entry_count(e);
    const float* host = reinterpret_cast<const float*>(st.pinned[half] + offset);
    if (e.kind == 3) {
        alphas_.assign(host, host + count);
        e.set = true;
        return;
    }
    ensure_arena(e.group);
    if (e.kind == 2) {
        SDMI_HIP(hipMemcpyAsync(*e.dst, host, count * sizeof(float), hipMemcpyHostToDevice, stream_));
    } else {
        float* stage = reinterpret_cast<float*>(st.dev[half] + offset);
        size_t n_stage = count;
        if (e.kind == 0 && e.dims[1] == 3) n_stage = count / 3 * 4;   // padded on the host by the caller of stage_commit
        SDMI_HIP(hipMemcpyAsync(stage, host, n_stage * sizeof(float), hipMemcpyHostToDevice, stream_));
        if (e.pre_scale != 1.f) SDMI_HIP(launch_scale_f32(stage, (long long)n_stage, e.pre_scale, stream_));
        hipError_t err;
        if (e.kind == 0) {
            const int cout = (int)e.dims[0], cin = e.dims[1] == 3 ? 4 : (int)e.dims[1], k = (int)e.dims[2];
            if (!(cin % 32 == 0 || (cin < 32 && cin % 4 == 0))) throw Error(SDMI_ERR_UNSUPPORTED, "conv Cin must be a multiple of 32, or < 32 and a multiple of 4");
            err = e.wdt ? launch_pack_conv_weight_bf16(stage, *e.dst, cout, cin, k, k, stream_)
                        : launch_pack_conv_weight(stage, *e.dst, cout, cin, k, k, stream_);
            if (err == hipSuccess && e.dst8) {
                const size_t kp = (size_t)((cin + 127) / 128 * 128) * k * k;
                if (!*e.dst8) {
                    void *q = nullptr, *sc = nullptr;
                    SDMI_HIP(hipMalloc(&q, (size_t)cout * kp));
                    weight_allocs_.push_back(q);
                    SDMI_HIP(hipMalloc(&sc, (size_t)cout * kp / 32));
                    weight_allocs_.push_back(sc);
                    *e.dst8 = reinterpret_cast<float*>(q);
                    *e.dsts = reinterpret_cast<float*>(sc);
                }
                err = launch_pack_conv_weight_fp8(stage, *e.dst8, *e.dsts, cout, cin, k, k, stream_);
            }
        } else {
            err = e.wdt ? launch_pack_linear_weight_bf16(stage, *e.dst, (int)e.dims[0], (int)e.dims[1], stream_)
                        : launch_pack_linear_weight(stage, *e.dst, (int)e.dims[0], (int)e.dims[1], stream_);
            if (err == hipSuccess && e.dst8) {
                const int cin = (int)e.dims[0], cout = (int)e.dims[1];
                const size_t kp = (size_t)(cin + 127) / 128 * 128;
                if (!*e.dst8) {
                    void *q = nullptr, *sc = nullptr;
                    SDMI_HIP(hipMalloc(&q, (size_t)cout * kp));
                    weight_allocs_.push_back(q);
                    SDMI_HIP(hipMalloc(&sc, (size_t)cout * kp / 32));
                    weight_allocs_.push_back(sc);
                    *e.dst8 = reinterpret_cast<float*>(q);
                    *e.dsts = reinterpret_cast<float*>(sc);
                }
                err = launch_pack_linear_weight_fp8(stage, *e.dst8, *e.dsts, cin, cout, stream_);
            }
        }
        SDMI_HIP(err);
        if (!e.wdt) {   // the bf16 planes of the packed fp32 rows (k_gemm3x.hip)
            const long long rows = e.kind == 0 ? e.dims[0] : e.dims[1];
            const long long K = e.kind == 0 ? (e.dims[1] == 3 ? 4 : e.dims[1]) * e.dims[2] * e.dims[3] : e.dims[0];
            void* planes = const_cast<void*>(split_planes(*e.dst));
            if (planes && K % 32 == 0) SDMI_HIP(launch_pack_split3(*e.dst, planes, rows, (int)K, stream_, b3_grouped(rows)));
        }
    }
    e.set = true;
    finalized_ = false;
}

// copies one tensor into the pinned ring (padding the RGB conv_in to 4 input channels) and commits it
void Engine::upload_weight(WeightEntry& e, const float* data) {
    const size_t count = entry_count(e);
    size_t off; int half;
    if (e.kind == 0 && e.dims[1] == 3) {
        const int cout = (int)e.dims[0], T = (int)(e.dims[2] * e.dims[3]);
        float* dst = reinterpret_cast<float*>(stage_reserve((size_t)cout * 4 * T * sizeof(float), &off, &half));
        std::memset(dst, 0, (size_t)cout * 4 * T * sizeof(float));
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < 3; ++c) std::memcpy(dst + ((size_t)o * 4 + c) * T, data + ((size_t)o * 3 + c) * T, T * sizeof(float));
    } else {
        char* dst = stage_reserve(count * sizeof(float), &off, &half);
        std::memcpy(dst, data, count * sizeof(float));
    }
    stage_commit(e, off, half);
}

// Module metadata the reference's loaders read next to the tensors (python/save.py:23-68): a norm's `eps` is honoured
// (groupnorm/load.rs:19, load.rs:load_layer_norm), everything else must equal what this engine is built for.
bool Engine::set_meta(const std::string& name, const float* values, size_t n) {
    auto it = meta_index_.find(name);
    if (it == meta_index_.end()) return false;
    MetaEntry& m = meta_[it->second];
    if ((int)n != m.n) throw Error(SDMI_ERR_WEIGHTS, "'" + name + "' holds " + std::to_string(n) + " values, expected " + std::to_string(m.n));
    if (m.store) {
        if (!(values[0] > 0.f) || values[0] > 1.f) throw Error(SDMI_ERR_WEIGHTS, "'" + name + "': eps out of range");
        *m.store = values[0];
        return true;
    }
    for (int i = 0; i < m.n; ++i)
        if (values[i] != m.expect[i]) {
            std::ostringstream os;
            os << "'" << name << "' = " << values[i] << " but this engine is built for " << m.expect[i]
               << " (the reference's loaders honour the file; a dump with different hyper-parameters needs a matching sdmi_config)";
            throw Error(SDMI_ERR_WEIGHTS, os.str());
        }
    return true;
}

void Engine::set_weight(const char* name, const float* data, int ndim, const int64_t* dims) {
    if (!name || !data || !dims) throw Error(SDMI_ERR_INVALID, "set_weight: null argument");
    if (meta_index_.count(name)) {
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)dims[i];
        set_meta(name, data, n);
        return;
    }
    auto it = entry_index_.find(name);
    if (it == entry_index_.end()) throw Error(SDMI_ERR_WEIGHTS, std::string("set_weight: unknown tensor '") + name + "'");
    WeightEntry& e = entries_[it->second];
    bool ok = ndim == e.ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = dims[i] == e.dims[i];
    if (!ok) {
        std::ostringstream os;
        os << "set_weight: '" << name << "' expects shape [";
        for (int i = 0; i < e.ndim; ++i) os << (i ? "," : "") << e.dims[i];
        os << "], got [";
        for (int i = 0; i < ndim; ++i) os << (i ? "," : "") << dims[i];
        os << "]";
        throw Error(SDMI_ERR_WEIGHTS, os.str());
    }
    SDMI_HIP(hipSetDevice(cfg_.device));
    upload_weight(e, data);   // data is copied into the pinned ring before this returns: the caller may free it
}

size_t Engine::packed_size(int groups) const {
    size_t n = 0;
    for (auto& e : entries_)
        if (groups & (1 << e.group)) n += entry_count(e);
    return n;
}

void Engine::load_weights_packed(const float* data, size_t n_floats, int groups) {
    if (!data) throw Error(SDMI_ERR_INVALID, "load_weights_packed: null pointer");
    if (groups <= 0 || groups > 7) throw Error(SDMI_ERR_INVALID, "load_weights_packed: groups is a bit mask of 1 (hot path), 2 (CLIP), 4 (VAE encoder)");
    if (n_floats != packed_size(groups)) throw Error(SDMI_ERR_WEIGHTS, "load_weights_packed: expected " + std::to_string(packed_size(groups)) + " floats, got " + std::to_string(n_floats));
    SDMI_HIP(hipSetDevice(cfg_.device));
    size_t off = 0;
    for (auto& e : entries_) {
        if (!(groups & (1 << e.group))) continue;
        upload_weight(e, data + off);
        off += entry_count(e);
    }
    SDMI_HIP(hipStreamSynchronize(stream_));
    stager_release();
}

// load_stable_diffusion_model_file (src/bin/sample/main.rs:27-34): the Burn record, read natively (mpk_reader.hpp for the
// assumed layout).  Tensors the configured model does not have (e.g. clip/... with clip_layers = 0) are skipped.
void Engine::load_weights_mpk(const char* path) {
    if (!path) throw Error(SDMI_ERR_INVALID, "load_weights_mpk: null path");
    SDMI_HIP(hipSetDevice(cfg_.device));
    MpkFile f(path);
    size_t used = 0;
    for (const MpkTensor& t : f.tensors()) {
        auto it = entry_index_.find(t.name);
        if (it == entry_index_.end()) continue;
        WeightEntry& e = entries_[it->second];
        bool ok = (int)t.shape.size() == e.ndim;
        for (int i = 0; ok && i < e.ndim; ++i) ok = t.shape[i] == e.dims[i];
        if (!ok) {
            std::ostringstream os;
            os << "load_weights_mpk: '" << t.name << "' has shape [";
            for (size_t i = 0; i < t.shape.size(); ++i) os << (i ? "," : "") << t.shape[i];
            os << "], the configured model expects [";
            for (int i = 0; i < e.ndim; ++i) os << (i ? "," : "") << e.dims[i];
            os << "]";
            throw Error(SDMI_ERR_WEIGHTS, os.str());
        }
        upload_weight(e, reinterpret_cast<const float*>(t.data));   // memcpy into the pinned ring: any alignment
        ++used;
    }
    if (!used) throw Error(SDMI_ERR_WEIGHTS, std::string("load_weights_mpk: no tensor of ") + path + " matches the configured model");
    SDMI_HIP(hipStreamSynchronize(stream_));
    stager_release();
}

void Engine::finalize_weights() {
    int total[3] = {0, 0, 0}, set[3] = {0, 0, 0};
    const WeightEntry* missing[3] = {nullptr, nullptr, nullptr};
    for (auto& e : entries_) {
        ++total[e.group];
        if (e.set) ++set[e.group];
        else if (!missing[e.group]) missing[e.group] = &e;
    }
    if (missing[0]) throw Error(SDMI_ERR_WEIGHTS, "finalize_weights: tensor '" + missing[0]->name + "' was never set");
    static const char* const kGroupName[3] = {"", "CLIP", "VAE encoder"};
    for (int g = 1; g < 3; ++g)
        if (set[g] && missing[g])
            throw Error(SDMI_ERR_WEIGHTS, std::string("finalize_weights: ") + kGroupName[g] + " weights are partially set; missing '" + missing[g]->name + "'");
    SDMI_HIP(hipStreamSynchronize(stream_));   // every packing kernel has run
    stager_release();
    clip_ready_ = total[1] > 0 && set[1] == total[1];
    enc_ready_ = total[2] > 0 && set[2] == total[2];
    finalized_ = true;
}

// npy-dump reader: src/model/load.rs:17-28 -- a 1-D float32 .npy whose first D
// values are the shape and whose remaining values are the row-major data.
// Returns the number of floats in the file; `sink(n)` supplies the destination for them.
template <class Sink>
static size_t read_npy_f32(const std::string& path, Sink&& sink) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(SDMI_ERR_IO, "cannot open " + path);
    char magic[8];
    f.read(magic, 8);
    if (!f || std::memcmp(magic, "\x93NUMPY", 6) != 0) throw Error(SDMI_ERR_IO, "not an npy file: " + path);
    uint32_t hlen = 0;
    if (magic[6] == 1) { uint16_t h16; f.read((char*)&h16, 2); hlen = h16; }
    else { f.read((char*)&hlen, 4); }
    std::string header(hlen, ' ');
    f.read(&header[0], hlen);
    if (header.find("'<f4'") == std::string::npos && header.find("\"<f4\"") == std::string::npos)
        throw Error(SDMI_ERR_IO, "npy dtype is not <f4: " + path);
    if (header.find("'fortran_order': True") != std::string::npos) throw Error(SDMI_ERR_IO, "fortran-order npy: " + path);
    const std::streampos start = f.tellg();
    f.seekg(0, std::ios::end);
    const size_t n = (size_t)(f.tellg() - start) / sizeof(float);
    f.seekg(start);
    float* dst = sink(n);
    f.read((char*)dst, (std::streamsize)(n * sizeof(float)));
    if (!f) throw Error(SDMI_ERR_IO, "short read: " + path);
    return n;
}

void Engine::load_weights_dir(const char* dir) {
    if (!dir) throw Error(SDMI_ERR_INVALID, "load_weights_dir: null path");
    SDMI_HIP(hipSetDevice(cfg_.device));
    // the CLIP subtree is read when it exists (load_stable_diffusion always has it, stablediffusion/load.rs:24)
    const bool have_clip = std::ifstream(std::string(dir) + "/clip/token_embedding/weight.npy").good();
    const bool have_enc = std::ifstream(std::string(dir) + "/autoencoder/encoder/conv_in/weight.npy").good();
    for (auto& m : meta_) {   // optional per-module metadata files
        const std::string path = std::string(dir) + "/" + m.name + ".npy";
        if (!std::ifstream(path).good()) continue;
        std::vector<float> raw;
        read_npy_f32(path, [&](size_t n) { raw.resize(n); return raw.data(); });
        // save_scalar writes [1.0, s]; save_tensor of a 2-vector writes [2.0, a, b] (python/save.py:6-15)
        if (raw.size() != (size_t)m.n + 1 || raw[0] != (float)m.n) throw Error(SDMI_ERR_WEIGHTS, "malformed metadata file " + path);
        set_meta(m.name, raw.data() + 1, (size_t)m.n);
    }
    for (auto& e : entries_) {
        if ((e.group == 1 && !have_clip) || (e.group == 2 && !have_enc)) continue;
        const std::string path = std::string(dir) + "/" + e.name + ".npy";
        const size_t count = entry_count(e);
        if (e.kind == 0 && e.dims[1] == 3) {   // rare (one tensor): through the padding path
            std::vector<float> raw;
            read_npy_f32(path, [&](size_t n) { raw.resize(n); return raw.data(); });
            if (raw.size() != count + (size_t)e.ndim) throw Error(SDMI_ERR_WEIGHTS, "shape prefix does not match payload in " + path);
            for (int i = 0; i < e.ndim; ++i)
                if ((int64_t)raw[i] != e.dims[i]) throw Error(SDMI_ERR_WEIGHTS, "unexpected shape in " + path);
            upload_weight(e, raw.data() + e.ndim);
            continue;
        }
        // the file's floats land in the pinned ring directly: [dims.., values..]; the values start ndim floats in
        size_t off = 0; int half = 0;
        char* base = nullptr;
        const size_t n = read_npy_f32(path, [&](size_t nf) {
            if (nf != count + (size_t)e.ndim) throw Error(SDMI_ERR_WEIGHTS, "shape prefix does not match payload in " + path);
            base = stage_reserve((nf + 64) * sizeof(float), &off, &half);
            // keep the VALUES 256-byte aligned for the H2D copy: the prefix sits right before them
            return reinterpret_cast<float*>(base + 256) - e.ndim;
        });
        (void)n;
        const float* pre = reinterpret_cast<const float*>(base + 256) - e.ndim;
        for (int i = 0; i < e.ndim; ++i)
            if ((int64_t)pre[i] != e.dims[i]) {
                std::ostringstream os;
                os << "unexpected shape in " << path << ": dim " << i << " is " << pre[i] << ", expected " << e.dims[i];
                throw Error(SDMI_ERR_WEIGHTS, os.str());
            }
        stage_commit(e, off + 256, half);
    }
    SDMI_HIP(hipStreamSynchronize(stream_));
    stager_release();
}

// =============================================================================
// primitive ops
// =============================================================================
Act Engine::new_act(int n, int h, int w, int c, int dt) {
    Act a; a.n = n; a.h = h; a.w = w; a.c = c; a.dt = dt < 0 ? edt() : dt;
    a.p = reinterpret_cast<float*>(pool_.alloc(a.bytes()));
    return a;
}
Act Engine::new_act3(int n, int h, int w, int c, int what) {
    if ((what & 2) && (c % 32 || bf16_)) throw Error(SDMI_ERR_STATE, "new_act3: planes need fp32 storage and c % 32 == 0");
    Act a; a.n = n; a.h = h; a.w = w; a.c = c; a.dt = 0;
    if (what & 1) a.p = reinterpret_cast<float*>(pool_.alloc(a.bytes()));
    if (what & 2) { a.p3 = pool_.alloc(a.bytes3()); a.ld3 = (c / 32) * 192; }
    return a;
}
void Engine::release(Act& a) {
    if (!a.view) {
        if (a.p) pool_.free(a.p);
        if (a.p3) pool_.free(a.p3);
    }
    a.p = nullptr; a.p3 = nullptr;
}

Act Engine::slice(const Act& parent, int c_off, int c) {
    if (c_off < 0 || c <= 0 || c_off + c > parent.c || parent.view) throw Error(SDMI_ERR_STATE, "slice: bad channel range");
    Act a = parent;
    a.p = parent.p ? adv(parent.p, c_off, parent.dt) : nullptr;
    if (parent.p3) {
        if (c_off % 32 || c % 32) throw Error(SDMI_ERR_STATE, "slice: plane tensors are cut at multiples of 32 channels");
        a.p3 = (char*)parent.p3 + (size_t)(c_off / 32) * 192;
    }
    a.c = c; a.ld = parent.stride(); a.view = true;
    return a;
}

void Engine::sync() { SDMI_HIP(hipStreamSynchronize(stream_)); }

void Engine::begin_call(bool dev_inputs) {
    SDMI_HIP(hipSetDevice(cfg_.device));
    n_kernels_ = 0; flops_ = 0;
    call_mark_ = pool_.serial();
    call_dev_ = dev_inputs;
    if (dev_inputs) {
        // the engine's stream is non-blocking: nothing orders it behind the stream that produced the caller's device
        // buffers unless we do
        if (has_user_stream_) {
            SDMI_HIP(hipEventRecord(ev_user_, user_stream_));
            SDMI_HIP(hipStreamWaitEvent(stream_, ev_user_, 0));
        } else {
            SDMI_HIP(hipDeviceSynchronize());
        }
    }
    SDMI_HIP(hipEventRecord(ev0_, stream_));
}
void Engine::end_call() {
    SDMI_HIP(hipEventRecord(ev1_, stream_));
    if (call_dev_ && has_user_stream_) SDMI_HIP(hipStreamWaitEvent(user_stream_, ev1_, 0));  // later work on the caller's stream sees the outputs
    SDMI_HIP(hipEventSynchronize(ev1_));
    float ms = 0;
    SDMI_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
    last_ms = ms; last_kernels = n_kernels_; last_flops = flops_;
}
void Engine::abort_call() noexcept {
    // a throw inside a forward pass leaves raw activations and the per-call UNet tables allocated: wait for what was
    // enqueued, then hand every block this call took back to the pool
    (void)hipStreamSynchronize(stream_);
    try {
        us_ = UNetState{};
        pool_.free_since(call_mark_);
    } catch (...) {}
}

void Engine::set_option(const std::string& key, const std::string& value) {
    if (key == "gemm_tile") opt_force_tile_ = (value == "auto") ? -1 : std::stoi(value);
    else if (key == "splitk") opt_force_splits_ = std::stoi(value);
    else if (key == "roctx") roctx_enable(std::stoi(value) != 0);
    else if (key == "fp8_convs") opt_fp8_convs_ = std::stoi(value);
    else if (key == "fp8_min_rows") opt_fp8_min_rows_ = std::stoi(value);
    else if (key == "fp8_linear") opt_fp8_linear_ = std::stoi(value);
    else if (key == "fp8_ops") opt_fp8_ops_ = std::stoi(value);
    else if (key == "op_resid") opt_op_resid_ = std::stoi(value);
    else if (key == "cfg_share") opt_cfg_share_ = std::stoi(value);
    else if (key == "attn_kv_splits") opt_attn_kv_splits_ = std::stoi(value);
    else if (key == "attn_kv_prefer8") opt_attn_kv_prefer8_ = std::stoi(value);
    else if (key == "b3_grouped") {
        if (!entries_.empty() && std::any_of(entries_.begin(), entries_.end(), [](const WeightEntry& w) { return w.set; }))
            throw Error(SDMI_ERR_STATE, "b3_grouped selects the layout the weight planes are packed in: set it before the first weight is loaded");
        opt_b3_grouped_ = std::stoi(value);
    }
    else if (key == "attn_pack_tail") opt_attn_pack_tail_ = (value == "default") ? 3 : std::stoi(value);
    else if (key == "gn32_min_wgs") opt_gn32_min_wgs_ = (opt_gn32_min_wgs_ & ~0xFFFF) | (std::stoi(value) & 0xFFFF);
    else if (key == "gn32_stats_min_wgs") opt_gn32_min_wgs_ = (opt_gn32_min_wgs_ & 0xFFFF) | ((std::stoi(value) + 1) << 16);   // the statistics pass cut differently from the apply pass (-1: the same)
    else if (key == "gn_target_wgs") gn_tune_.target_wgs = std::stoi(value);
    else if (key == "gn_max_threads") gn_tune_.max_threads = std::stoi(value);
    else if (key == "gn_unroll") gn_tune_.unroll = std::stoi(value);
    else if (key == "fp8_tile") opt_fp8_tile_ = (value == "auto") ? -1 : std::stoi(value);
    else if (key == "resid_acc") opt_resid_acc_ = std::stoi(value);
    else if (key == "attn_bf16") opt_attn_bf16_ = std::stoi(value);
    else if (key == "attn_bf16_variant") opt_attn_bf16_variant_ = (value == "default") ? kAttnBf16VariantDefault : std::stoi(value, nullptr, 0);
    else if (key == "attn_split") opt_attn_split_ = std::stoi(value);
    else if (key == "gemm_bf16x") opt_gemm_bf16x_ = std::stoi(value);
    else if (key == "gemm_x32") opt_gemm_x32_ = std::stoi(value);
    else if (key == "gemm_f32s") opt_gemm_f32s_ = std::stoi(value);
    else if (key == "bench_cold") opt_bench_cold_ = std::stoi(value);
    else if (key == "gemm_probe") opt_gemm_probe_ = std::stoi(value);
    else if (key == "conv3_reuse") opt_conv3_reuse_ = std::stoi(value);
    else if (key == "gemm_planes") opt_gemm_planes_ = (value == "default") ? kGemmPlanesDefault : std::stoi(value);
    else if (key == "gemm3x_variant") opt_gemm3x_variant_ = (value == "default") ? kGemm3xVariantDefault : std::stoi(value);
    else if (key == "gemm_bf16x_variant") opt_gemm_bf16x_variant_ = (value == "default") ? kGemmBf16xVariantDefault : std::stoi(value);
    else if (key == "geglu_fuse") opt_geglu_fuse_ = std::stoi(value);
    else if (key == "record_shapes") { record_shapes_ = std::stoi(value) != 0; if (record_shapes_) { shape_counts_.clear(); choice_counts_.clear(); } }
    else if (key == "dump_shapes") {
        std::ofstream f(value);
        if (!f) throw Error(SDMI_ERR_IO, "dump_shapes: cannot write " + value);
        for (auto& kv : shape_counts_) f << kv.first << " " << kv.second << "\n";
    }
    else if (key == "dump_choices") {
        std::ofstream f(value);
        if (!f) throw Error(SDMI_ERR_IO, "dump_choices: cannot write " + value);
        for (auto& kv : choice_counts_) f << kv.first << " x" << kv.second << "\n";
    }
    else if (key == "profile") { prof_flush(); profiling_ = std::stoi(value) != 0; prof_tagging_ = std::stoi(value) >= 2; if (profiling_) prof_calibrate(); }
    else if (key == "dump_profile_tags") {   // profile=2: "ms launches flops bytes<TAB>tag" per line
        prof_flush();
        std::ofstream f(value);
        if (!f) throw Error(SDMI_ERR_IO, "dump_profile_tags: cannot write " + value);
        for (auto& kv : prof_tags_) f << kv.second.ms << " " << kv.second.launches << " " << kv.second.flops << " " << kv.second.bytes << "\t" << kv.first << "\n";
    }
    else if (key == "profile_reset") prof_reset();
    else if (key == "tune" || key == "tune_bf16") {
        // "M,N,K=cfg,splits" (tune_bf16: cfg 100 + x selects a k_gemm_bf16x.hip tile)
        const size_t eq = value.find('=');
        if (eq == std::string::npos) throw Error(SDMI_ERR_INVALID, "tune expects M,N,K=cfg,splits");
        const bool b16 = key == "tune_bf16";
        TileChoice tc{0, 1};
        if (std::sscanf(value.c_str() + eq + 1, "%d,%d", &tc.cfg, &tc.splits) != 2 || tc.cfg < 0 || tc.splits < 1 ||
            !(tc.cfg < kNumGemmTiles || (tc.cfg >= 100 && tc.cfg < 100 + (b16 ? kNumGemmTilesXB : kNumGemmTilesX)) || (!b16 && tc.cfg >= 200 && tc.cfg < 200 + kNumGemmTilesS) ||
              (!b16 && tc.cfg >= 300 && tc.cfg < 300 + kNumGemmTilesP)))
            throw Error(SDMI_ERR_INVALID, "tune: bad value");
        (b16 ? tuned_bf16_ : (tc.cfg >= 300 ? tuned_p_ : tuned_))[value.substr(0, eq)] = tc;   // plane tiles (300 + x) have their own table: what a GEMM whose input arrives as planes chooses from
    } else if (key == "tune_clear") { tuned_.clear(); tuned_bf16_.clear(); tuned_mfma_.clear(); tuned_p_.clear(); }
    else throw Error(SDMI_ERR_INVALID, "unknown option '" + key + "'");
}

// Heuristic tile / split-K choice: minimise (work per CU after round-robin
// placement) / (tile efficiency) + split-K slab traffic.  Overridden per shape by
// measured entries ("tune" option, see tools/autotune.py).  The K-reduction order
// depends only on (M,N,K), so a sample's result is independent of where it sits
// in the batch only for equal M; see DESIGN.md "Determinism".
TileChoice Engine::choose_tile(int M, int N, int kt_total, bool allow_x, bool allow_s) const {
    static const double eff[kNumGemmTiles] = {0.85, 0.75, 0.60, 0.90, 0.75, 0.85, 0.75, 0.85, 0.65, 0.75};
    static const int split_opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48};
    const int n_cu = 256;
    double best = 1e300;
    TileChoice bc{0, 1};
    for (int c = 0; c < kNumGemmTiles; ++c) {
        const int bm = gemm_tile_info(c).bm, bn = gemm_tile_info(c).bn;
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        for (int s : split_opts) {
            if (s > 1 && kt_total / s < 4) break;
            const int kt_per = (kt_total + s - 1) / s;
            const long long wgs = tiles * ((kt_total + kt_per - 1) / kt_per);
            const double per_cu = (double)((wgs + n_cu - 1) / n_cu);
            // cycles: one k tile of a bm x bn block = bm*bn*32*2 flop at 256 flop/clk/CU
            double t = per_cu * (double)bm * bn * kt_per * 64.0 / 256.0 / eff[c];
            t += 3000.0 * per_cu;  // prologue / epilogue per workgroup
            if (s > 1) t += 8000.0 + (double)M * N * 4.0 * (s + 1) / (5.0e12 / 2.4e9);  // reduce launch + slab traffic
            if (t < best) { best = t; bc = {c, s}; }
        }
    }
    if (allow_x && opt_gemm_x32_) {
        // k_gemm2x.hip: 8 waves, one workgroup per CU; ~0.9 of the matrix rate in the k loop, but the DMA prologue and the
        // output tile's store are not hidden by a neighbour
        static const double eff_x[kNumGemmTilesX] = {0.90, 0.90, 0.88, 0.90};
        for (int c = 0; c < kNumGemmTilesX; ++c) {
            const int bm = gemm_tile_info_x(c).bm, bn = gemm_tile_info_x(c).bn;
            const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
            for (int s : split_opts) {
                if (s > 1 && kt_total / s < 4) break;
                const int kt_per = (kt_total + s - 1) / s;
                const long long wgs = tiles * ((kt_total + kt_per - 1) / kt_per);
                const double per_cu = (double)((wgs + n_cu - 1) / n_cu);
                double t = per_cu * ((double)bm * bn * kt_per * 64.0 / 256.0 / eff_x[c] + 8000.0 + bm * bn * 4.0 / 10.0);
                if (s > 1) t += 8000.0 + (double)M * N * 4.0 * (s + 1) / (5.0e12 / 2.4e9);
                if (t < best) { best = t; bc = {100 + c, s}; }
            }
        }
    }
    if (allow_s && opt_gemm_f32s_) {
        // k_gemm3x.hip: six bf16 MFMAs per 16x16x32 block = 96 cycles/SIMD against 256 on the fp32 pipe; the efficiencies
        // are measured ones (tools/autotune.py), the per-workgroup constant covers the DMA prologue and the epilogue
        static const double eff_s[kNumGemmTilesS] = {0.55, 0.55, 0.52, 0.52, 0.46, 0.44};
        for (int c = 0; c < kNumGemmTilesS; ++c) {
            const int bm = gemm_tile_info_s(c).bm, bn = gemm_tile_info_s(c).bn;
            const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
            for (int s : split_opts) {
                if (s > 1 && kt_total / s < 4) break;
                const int kt_per = (kt_total + s - 1) / s;
                const long long wgs = tiles * ((kt_total + kt_per - 1) / kt_per);
                const double per_cu = (double)((wgs + n_cu - 1) / n_cu);
                double t = per_cu * ((double)bm * bn * kt_per * 384.0 / 4096.0 / eff_s[c] + 8000.0 + bm * bn * 4.0 / 10.0);
                if (s > 1) t += 8000.0 + (double)M * N * 4.0 * (s + 1) / (5.0e12 / 2.4e9);
                if (t < best) { best = t; bc = {200 + c, s}; }
            }
        }
    }
    return bc;
}

// precision = 1: the 4-wave tiles of k_gemm_bf16.hip (cfg 0..9) against the 8-wave LDS-DMA tiles of
// k_gemm_bf16x.hip (cfg 100 + x).  Cycles per CU at 4096 bf16 flop/clk/CU; the efficiencies are measured
// ones (tools/autotune.py --precision bf16), the per-workgroup constants cover prologue DMA latency + epilogue.
TileChoice Engine::choose_tile_bf16(int M, int N, int kt_total) const {
    static const double eff_old[kNumGemmTiles] = {0.31, 0.22, 0.16, 0.22, 0.20, 0.20, 0.22, 0.26, 0.15, 0.28};
    static const double eff_x[kNumGemmTilesX] = {0.54, 0.46, 0.38, 0.48};
    static const int split_opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48};
    const int n_cu = 256;
    double best = 1e300;
    TileChoice bc{0, 1};
    auto consider = [&](int cfg, int bm, int bn, double eff, double wg_overhead) {
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        for (int s : split_opts) {
            if (s > 1 && kt_total / s < 4) break;
            const int kt_per = (kt_total + s - 1) / s;
            const long long wgs = tiles * ((kt_total + kt_per - 1) / kt_per);
            const double per_cu = (double)((wgs + n_cu - 1) / n_cu);
            double t = per_cu * ((double)bm * bn * kt_per * 128.0 / 4096.0 / eff + wg_overhead);
            if (s > 1) t += 8000.0 + (double)M * N * 4.0 * (s + 1) / (5.0e12 / 2.4e9);
            if (t < best) { best = t; bc = {cfg, s}; }
        }
    };
    for (int c = 0; c < kNumGemmTiles; ++c) consider(c, gemm_tile_info(c).bm, gemm_tile_info(c).bn, eff_old[c], 3000.0);
    if (opt_gemm_bf16x_)
        for (int c = 0; c < kNumGemmTilesX; ++c) {
            // one workgroup per CU (144 KB of LDS): the DMA prologue and the output tile's store are not hidden by a neighbour
            const int bm = gemm_tile_info_x(c).bm, bn = gemm_tile_info_x(c).bn;
            consider(100 + c, bm, bn, eff_x[c], 6000.0 + bm * bn * 2.0 / 20.0);
        }
    return bc;
}

// k_gemm3p.hip tiles (300 + x): what a GEMM whose activations arrive as planes chooses from when its shape is not in the measured table
// (tuning/gfx950_fp32_planes.txt).  Same cost form as choose_tile's split branch; efficiencies from tools/autotune.py --families p.
TileChoice Engine::choose_tile_p(int M, int N, int kt_total, bool even_ni_only) const {
    static const double eff_p[kNumGemmTilesP] = {0.62, 0.60, 0.60, 0.52, 0.50, 0.34, 0.40, 0.50, 0.40};
    static const int split_opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48};
    const int n_cu = 256;
    double best = 1e300;
    TileChoice bc{304, 1};
    for (int c = 0; c < kNumGemmTilesP; ++c) {
        if (even_ni_only && (c == 0 || c == 3 || c == 7)) continue;
        const int bm = gemm_tile_info_p(c).bm, bn = gemm_tile_info_p(c).bn;
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        for (int s : split_opts) {
            if (s > 1 && kt_total / s < 4) break;
            const int kt_per = (kt_total + s - 1) / s;
            const long long wgs = tiles * ((kt_total + kt_per - 1) / kt_per);
            const double per_cu = (double)((wgs + n_cu - 1) / n_cu);
            double t = per_cu * ((double)bm * bn * kt_per * 384.0 / 4096.0 / eff_p[c] + 8000.0 + bm * bn * 4.0 / 10.0);
            if (s > 1) t += 8000.0 + (double)M * N * 4.0 * (s + 1) / (5.0e12 / 2.4e9);
            if (t < best) { best = t; bc = {300 + c, s}; }
        }
    }
    return bc;
}

void Engine::launch_gemm(ConvGemm& p, int in_dt, int force_cfg, int force_splits) {
    const int kt_elems = in_dt ? 64 : 32;  // a k tile is 128 bytes of K per row in both storage types
    p.kt_total = (p.K + kt_elems - 1) / kt_elems;
    if (in_dt && (p.Cin % 64)) throw Error(SDMI_ERR_UNSUPPORTED, "bf16 GEMM: K slices must be multiples of 64");
    if (record_shapes_) {
        char sk[96];
        std::snprintf(sk, sizeof sk, "%d,%d,%d,%d,%d,%d,%d,%d", p.NB, p.Cin, p.Hs, p.Ws, p.N, p.KH, p.stride, p.ups);
        ++shape_counts_[sk];
    }
    auto tile_info = [&](int cfg) -> const GemmTileInfo& {
        if (in_dt) return cfg >= 100 ? gemm_tile_info_xb(cfg - 100) : gemm_tile_info(cfg);
        return cfg >= 300 ? gemm_tile_info_p(cfg - 300) : cfg >= 200 ? gemm_tile_info_s(cfg - 200) : cfg >= 100 ? gemm_tile_info_x(cfg - 100) : gemm_tile_info(cfg);
    };
    TileChoice tc;
    char key[64];
    std::snprintf(key, sizeof key, "%d,%d,%d", p.M, p.N, p.K);
    const auto& table = in_dt ? tuned_bf16_ : tuned_;  // measured per storage type (tuning/gfx950_{fp32,bf16}.txt)
    auto it = table.find(key);
    const bool x32_ok = !in_dt && p.CS == 32 && p.Cin % 32 == 0 && p.out_mode == 0;   // what k_gemm2x.hip handles
    p.Bt3 = in_dt ? nullptr : split_planes(p.Bt);
    p.b3_grouped = b3_grouped((long long)p.N * (p.geglu ? 2 : 1)) ? 1 : 0;   // the layout the planes of a weight with that many rows were packed in
    p.variant = in_dt ? opt_gemm_bf16x_variant_ : opt_gemm3x_variant_;
    // k_gemm3x.hip: the same layers, when the weight has its bf16 planes (weights in the arenas; not e.g. the K / V operands of
    // the unfused VAE attention) and the 32-bit piece offsets reach
    const bool s_ok = x32_ok && p.Bt3 && (unsigned long long)p.N * (p.geglu ? 2 : 1) * (unsigned long long)p.kt_total * 192ull < 0xFFFFFF00ull;
    // k_gemm3p.hip (300 + x): the same layers with the activations as planes too -- written by their producer (p.A3) or, for a tensor that
    // arrives as fp32, by split3_rows_kernel right here
    const bool p_ok = s_ok && (unsigned long long)p.NB * p.Hs * p.Ws * (unsigned long long)(p.A3 ? p.a3_ld : p.Cin * 6) < 0xFFFFFF00ull;
    // the activations arrive as planes (their producer wrote them): the GEMM runs on a plane tile -- from the plane table or the cost model
    const bool from_planes = !in_dt && p.A3 != nullptr;
    if (from_planes && !p_ok)
        throw Error(s_ok ? SDMI_ERR_UNSUPPORTED : SDMI_ERR_STATE,
                    s_ok ? "fp32 GEMM: an activation tensor stored as bf16 planes (6 bytes per element) reaches 4 GiB (32-bit piece offsets): lower the batch (at 64x64x960 the CFG "
                           "batch 2n must stay <= 182) or set option gemm_planes=0"
                         : "gemm: activation planes given for a layer the plane kernel does not take");
    if (!from_planes && !p.A) throw Error(SDMI_ERR_STATE, "gemm: no activations");
    auto usable = [&](int cfg) { return cfg < 100 || (in_dt ? opt_gemm_bf16x_ != 0 : (cfg >= 300 ? p_ok : cfg >= 200 ? (opt_gemm_f32s_ != 0 && s_ok) : (opt_gemm_x32_ != 0 && x32_ok))); };
    const auto it2 = in_dt ? tuned_mfma_.end() : tuned_mfma_.find(key);   // the table measured without the split kernels
    if (from_planes) {
        const auto itp = tuned_p_.find(key);
        const bool even = p.geglu != 0;
        if (itp != tuned_p_.end() && !(even && (itp->second.cfg == 300 || itp->second.cfg == 303 || itp->second.cfg == 307))) tc = itp->second;
        else tc = choose_tile_p(p.M, p.N, p.kt_total, even);
    }
    else if (it != table.end() && usable(it->second.cfg)) tc = it->second;
    else if (it2 != tuned_mfma_.end() && usable(it2->second.cfg)) tc = it2->second;
    else tc = in_dt ? choose_tile_bf16(p.M, p.N, p.kt_total) : choose_tile(p.M, p.N, p.kt_total, x32_ok, s_ok);
    bool tile_forced = false;   // by option gemm_tile or by the caller: the launch then runs exactly that tile (no upgrade to the kernel-row form below)
    if (opt_force_tile_ >= 0 && (from_planes ? opt_force_tile_ >= 300 : (opt_force_tile_ < 100 || in_dt || (opt_force_tile_ >= 300 ? p_ok : opt_force_tile_ >= 200 ? s_ok : x32_ok)))) { tc.cfg = opt_force_tile_; tile_forced = true; }  // 100+ / 200+: large-tile kernels, where applicable
    if (opt_force_splits_ > 0) tc.splits = opt_force_splits_;
    if (force_cfg >= 0) { tc.cfg = force_cfg; tile_forced = true; }
    // gemm_planes = 2 (A/B switch, tests): every launch that chose a k_gemm3x.hip tile runs on the k_gemm3p.hip tile nearest in shape, its
    // fp32 activations converted by split3_rows_kernel in front of it
    if (!in_dt && !from_planes && p_ok && opt_gemm_planes_ == 2 && tc.cfg >= 200 && tc.cfg < 200 + kNumGemmTilesS) {
        static const int kSplitToP[kNumGemmTilesS] = {300, 303, 301, 302, 303, 304};
        const int c = kSplitToP[tc.cfg - 200];
        if (!p.geglu || c == 301 || c == 302 || c == 304) tc.cfg = c;   // (even fragment counts only for the GEGLU epilogue)
    }
    if (from_planes && tc.cfg < 300) throw Error(SDMI_ERR_STATE, "gemm: activation planes need a plane tile (300 + x)");
    if (force_splits > 0) tc.splits = force_splits;
    if (!in_dt && p.out_mode == 2) tc.splits = 1;  // fp32 kernel emitting bf16: no split-K path
    int splits = std::max(1, std::min(tc.splits, p.kt_total));
    p.kt_per_split = (p.kt_total + splits - 1) / splits;
    splits = (p.kt_total + p.kt_per_split - 1) / p.kt_per_split;
    p.splits = splits;
    const double flops = 2.0 * p.M * (double)p.N * p.K * (p.geglu ? 2.0 : 1.0);
    // raw buffer loads: the range check needs 32-bit extents
    const unsigned long long es = in_dt ? 2ull : 4ull;
    const unsigned long long a_ext = ((unsigned long long)p.NB * p.Hs * p.Ws - 1) * (unsigned long long)p.a_ld * es + (unsigned long long)p.Cin * es;
    const unsigned long long b_ext = ((unsigned long long)p.N * (p.geglu ? 2 : 1) - 1) * (unsigned long long)p.b_ld * es + (unsigned long long)p.K * es;
    p.zero_page = zero_page_;
    // bf16 3x3 / stride-1 convolutions on the 256 x 320 / 256 x 256 tiles: the form that stages a kernel row's activations once for its three taps (k_gemm_bf16t.hip)
    if (in_dt && opt_conv3_reuse_ && !tile_forced && (tc.cfg == 100 || tc.cfg == 101) && conv_gemm_bf16t_supported(p)) tc.cfg += kNumGemmTilesX;
    if (record_shapes_) {   // which kernel / tile / split-K each (M, N, K) got: option dump_choices
        char ck[128];
        std::snprintf(ck, sizeof ck, "%d,%d,%d k%d s%d u%d W%d cfg=%d splits=%d%s", p.M, p.N, p.K, p.KH, p.stride, p.ups, p.Ws, tc.cfg, splits, p.Bt3 || in_dt ? "" : " (no planes)");
        ++choice_counts_[ck];
    }
    if (tc.cfg >= 300 ? (in_dt || tc.cfg - 300 >= kNumGemmTilesP || !p_ok)
        : tc.cfg >= 200 ? (in_dt || tc.cfg - 200 >= kNumGemmTilesS || !s_ok) : (tc.cfg >= 100 && (tc.cfg - 100 >= (in_dt ? kNumGemmTilesXB : kNumGemmTilesX) || (!in_dt && !x32_ok))))
        throw Error(SDMI_ERR_INVALID, "gemm: large-tile kernel index out of range or not applicable to this layer");
    if (a_ext >= 0xFFFFFFE0ull || b_ext >= 0xFFFFFFE0ull) throw Error(SDMI_ERR_UNSUPPORTED, "GEMM: operand larger than 4 GiB (the buffer-load range check needs 32-bit extents)");
    p.a_bytes = (unsigned)a_ext;
    p.b_bytes = (unsigned)b_ext;
    // the output as planes (p.C3): written by the epilogue of the split / plane kernels and by the split-K reduce kernel on their 16-byte
    // path; otherwise (old kernels, odd strides, GEGLU epilogue) converted from an fp32 result right behind the launch
    void* const c3_want = in_dt ? nullptr : p.C3;
    const int ldc3_want = p.ldc3;
    const bool vec_out = (p.N % 4 == 0) && (!p.C || p.ldc % 4 == 0) && (!p.resid || p.ldr % 4 == 0);
    const bool c3_native = c3_want && tc.cfg >= 200 && vec_out && (!p.geglu || (p.geglu == 2 && tc.cfg >= 300));
    std::unique_ptr<Buf> c_tmp;
    if (!c3_native) {
        p.C3 = nullptr;
        if (c3_want && !p.C) {
            c_tmp.reset(new Buf(this, (size_t)p.M * p.N * sizeof(float)));
            p.C = c_tmp->f(); p.ldc = p.N;
        }
    }
    if (!p.C) p.ldc = p.N;
    std::unique_ptr<Buf> a3_tmp;
    if (tc.cfg >= 300 && !p.A3) {   // the source is fp32: split it once for this launch (a producer that writes planes itself saves this pass)
        const long long rows = (long long)p.NB * p.Hs * p.Ws;
        p.a3_ld = (p.Cin / 32) * 192;
        a3_tmp.reset(new Buf(this, (size_t)rows * p.a3_ld));
        ProfScope ps(this, PC_SPLIT_ROWS, 0, (double)rows * p.Cin * 10.0);
        SDMI_HIP(launch_split3_rows(p.A, a3_tmp->p, rows, p.Cin, p.a_ld, p.a3_ld, stream_));
        count_kernel();
        p.A3 = a3_tmp->p;
    }
    p.probe = probe_buf_;
    auto launch = [&](const ConvGemm& q) {
        if (tc.cfg >= 300) return launch_conv_gemm3p(q, tc.cfg - 300, stream_);
        if (in_dt && tc.cfg >= 100) return launch_conv_gemm_bf16_large(q, tc.cfg - 100, stream_);
        if (tc.cfg >= 200) return launch_conv_gemm3x(q, tc.cfg - 200, stream_);
        if (tc.cfg >= 100) return launch_conv_gemm2x(q, tc.cfg - 100, stream_);
        if (in_dt) return launch_conv_gemm_bf16(q, tc.cfg, stream_);
        return launch_conv_gemm2(q, tc.cfg, stream_);
    };
    p.slabs = nullptr;
    // round 6: the large-tile bf16 kernels take the residual as the accumulators' initial value (k_gemm_bf16_epi.hpp gemm_acc_init_bf16) where their 8-byte loads apply
    // (option resid_acc: bit 0 = the residual, bit 1 = bias + time-embedding row; launches without split-K only -- the split-K combine adds them otherwise)
    p.resid_acc = 0;
    if (in_dt && tc.cfg >= 100 && splits == 1 && (p.N % 8) == 0 && (p.ldc % 8) == 0) {
        if ((opt_resid_acc_ & 1) && p.resid && !p.geglu && (p.ldr % 4) == 0) p.resid_acc |= 1;
        if ((opt_resid_acc_ & 2) && (p.bias || p.rowvec) && (p.rowvec_stride % 4) == 0) p.resid_acc |= 2;
    }
    const int pc = (!in_dt && tc.cfg >= 200) ? PC_CONV_SPLIT : PC_CONV_GEMM;   // k_gemm3x.hip launches are timed as their own class
    // ALGORITHMIC bytes of the launch in the formats the tensors are stored in: the source activations once, the weights once, the result once (bf16 2 B, fp32 4 B,
    // planes 6 B per element; split-K slabs and im2col / tile re-reads are not algorithmic) -- what the PMC byte counters of profiles/pmc_summary.json are held against
    const double a_es = in_dt ? 2.0 : (tc.cfg >= 300 ? 6.0 : 4.0), w_es = in_dt ? 2.0 : (tc.cfg >= 200 ? 6.0 : 4.0);
    const double c_es = in_dt ? (p.out_mode == 1 ? 4.0 : 2.0) : (p.out_mode == 2 ? 2.0 : ((p.C ? 4.0 : 0.0) + (c3_native ? 6.0 : 0.0)));
    const double gemm_bytes = (double)p.NB * p.Hs * p.Ws * p.Cin * a_es + (double)p.N * (p.geglu ? 2.0 : 1.0) * p.K * w_es + (double)p.M * p.N * c_es;
    if (splits == 1) {
        p.slab_stride = 0;
        ProfScope ps(this, pc, flops, gemm_bytes);
        ps.set_tag("gemm %d,%d,%d k%d%s%s%s cfg=%d splits=1", p.M, p.N, p.K, p.KH, p.geglu ? " geglu" : "", p.resid ? " resid" : "", p.rowvec ? " rowvec" : "", tc.cfg);
        SDMI_HIP(launch(p));
        count_kernel(flops);
    } else {
        p.slab_stride = (long long)p.M * p.N;
        Buf slab(this, (size_t)splits * p.slab_stride * sizeof(float));
        p.slabs = slab.f();
        {
            ProfScope ps(this, pc, flops, gemm_bytes);
            ps.set_tag("gemm %d,%d,%d k%d%s cfg=%d splits=%d", p.M, p.N, p.K, p.KH, p.geglu ? " geglu" : "", tc.cfg, splits);
            SDMI_HIP(launch(p));
        }
        count_kernel(flops);
        {
            ProfScope ps(this, PC_SPLITK_REDUCE, 0, (double)(splits + 1) * p.slab_stride * 4.0);
            ps.set_tag("reduce %d,%d,%d k%d cfg=%d splits=%d", p.M, p.N, p.K, p.KH, tc.cfg, splits);
            if (in_dt) SDMI_HIP(launch_splitk_reduce_bf16(p, stream_));
            else SDMI_HIP(launch_splitk_reduce(p, stream_));
            count_kernel();
        }
    }
    if (c3_want && !c3_native) {
        ProfScope ps(this, PC_SPLIT_ROWS, 0, (double)p.M * p.N * 10.0);
        SDMI_HIP(launch_split3_rows(p.C, c3_want, p.M, p.N, p.ldc, ldc3_want, stream_));
        count_kernel();
    }
}

void Engine::conv(const ConvW& w, const Act& x, Act& y, int stride, int ups, const float* rowvec, int rowvec_stride,
                  const Act* resid, bool pad_br) {
    if (x.c != w.cin) throw Error(SDMI_ERR_INVALID, "conv: input channels mismatch");
    // pad_br: rows / columns past the bottom / right edge read as zero through the kernels' range check, so the
    // asymmetric padding is pad = 0 plus one more output row / column than a symmetric pad-0 conv has
    const int pad = pad_br ? 0 : (w.k == 3 ? 1 : 0);
    const int hin = x.h << ups, win = x.w << ups;
    const int extra = pad_br ? 1 : 0;
    const int ho = (hin + 2 * pad + extra - w.k) / stride + 1, wo = (win + 2 * pad + extra - w.k) / stride + 1;
    if (y.n != x.n || y.h != ho || y.w != wo || y.c != w.cout) throw Error(SDMI_ERR_INVALID, "conv: output shape mismatch");
    ConvGemm p{};
    if (resid && (resid->rows() != y.rows() || resid->c != y.c || resid->dt != y.dt)) throw Error(SDMI_ERR_STATE, "conv: residual shape / type mismatch");
    p.A = x.p; p.Bt = w.bt; p.C = y.p; p.bias = w.bias; p.rowvec = rowvec; p.resid = resid ? resid->p : nullptr;
    if (resid && !resid->p) throw Error(SDMI_ERR_STATE, "conv: the residual must exist as fp32");
    if (!x.dt && plane_gemm(w.cin, w.cout) && split_planes(w.bt)) { p.A3 = x.p3; p.a3_ld = x.ld3; }   // planes in, where the plane kernel takes the layer
    if (!x.p && !p.A3) throw Error(SDMI_ERR_STATE, "conv: the input exists only as planes, which this layer cannot read");
    p.C3 = y.p3; p.ldc3 = y.ld3;
    p.M = x.n * ho * wo; p.N = w.cout; p.K = w.cin * w.k * w.k;
    p.NB = x.n; p.Hs = x.h; p.Ws = x.w; p.Cin = w.cin; p.Ho = ho; p.Wo = wo;
    p.KH = w.k; p.KW = w.k; p.stride = stride; p.pad = pad; p.ups = ups;
    p.ldc = y.stride(); p.ldr = resid ? resid->stride() : y.stride(); p.a_ld = x.stride(); p.b_ld = p.K; p.rowvec_stride = rowvec_stride;
    p.CS = std::min(32, w.cin);
    if (!y.p && !y.p3) throw Error(SDMI_ERR_STATE, "conv: no output buffer");
    if (x.dt != w.dt) throw Error(SDMI_ERR_STATE, "conv: activation / weight storage types disagree");
    p.out_mode = x.dt ? (y.dt ? 0 : 1) : (y.dt ? 2 : 0);
    if (resid && !x.dt && y.dt) throw Error(SDMI_ERR_STATE, "conv: residual not supported on the fp32->bf16 layers");
    launch_gemm(p, x.dt);
}

void Engine::gemm(const float* A, int a_rows, const float* bt, const float* bias, int cin, int cout, float* C, int ldc,
                  const float* resid, int ldr, int dt, int out_mode, const void* A3, void* C3) {
    if (dt < 0) dt = edt();
    if (cin % 32) throw Error(SDMI_ERR_UNSUPPORTED, "linear: in_features must be a multiple of 32");
    ConvGemm p{};
    p.A = A; p.Bt = bt; p.C = C; p.bias = bias; p.resid = resid;
    p.A3 = A3; p.a3_ld = (cin / 32) * 192;          // dense planes in ...
    p.C3 = C3; p.ldc3 = (cout / 32) * 192;          // ... and out
    if (C3 && (cout % 32)) throw Error(SDMI_ERR_STATE, "linear: plane output needs out_features % 32 == 0");
    p.M = a_rows; p.N = cout; p.K = cin;
    p.NB = 1; p.Hs = 1; p.Ws = a_rows; p.Cin = cin; p.Ho = 1; p.Wo = a_rows;
    p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.ups = 0;
    p.ldc = ldc; p.ldr = ldr; p.a_ld = cin; p.b_ld = cin; p.rowvec_stride = 0; p.CS = 32;
    p.out_mode = out_mode;
    launch_gemm(p, dt);
}

void Engine::gemm_geglu(const float* x, long long rows, const float* bt, const float* bias, int cin, int hidden, float* out, int dt,
                        const void* x3, void* out3) {
    if (dt < 0) dt = edt();
    if (x3 && (out3 || out) && !dt && opt_geglu_fuse_ && hidden % 32 == 0 && cin % 32 == 0 && split_planes(bt)) {
        // round 5: the gate in the plane GEMM's epilogue -- value / gate rows split by WAVE column, so the tiles with an odd fragment count per wave (256 x 160:
        // the batch-1 model's) qualify.  The tile is the one the unfused projection [rows, 2 hidden] would take (same tile count: 80 outputs = 160 weight rows per
        // tile); it must run without split-K and have an even number of wave columns (not 128 x 64).
        const int kt_total = (cin + 31) / 32;
        char key[64];
        std::snprintf(key, sizeof key, "%lld,%d,%d", rows, 2 * hidden, cin);
        const auto itp = tuned_p_.find(key);
        const TileChoice tc = itp != tuned_p_.end() ? itp->second : choose_tile_p((int)rows, 2 * hidden, kt_total, false);
        const int force = opt_force_tile_ >= 300 ? opt_force_tile_ : tc.cfg;
        if (force >= 300 && force != 308 && (tc.splits == 1 || opt_force_tile_ >= 300) && opt_force_splits_ <= 1) {
            ConvGemm p{};
            p.A = x; p.Bt = bt; p.C = out; p.bias = bias;
            p.A3 = x3; p.a3_ld = (cin / 32) * 192;
            p.C3 = out3; p.ldc3 = (hidden / 32) * 192;
            p.M = (int)rows; p.N = hidden; p.K = cin;
            p.NB = 1; p.Hs = 1; p.Ws = (int)rows; p.Cin = cin; p.Ho = 1; p.Wo = (int)rows;
            p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.ups = 0;
            p.ldc = hidden; p.ldr = hidden; p.a_ld = cin; p.b_ld = cin; p.rowvec_stride = 0; p.CS = 32;
            p.out_mode = 0;
            p.geglu = 2;
            launch_gemm(p, 0, force, 1);
            return;
        }
    }
    if (x3 || out3) {   // plane form: projection on a plane tile, gate kernel writing planes (the MLP's second Linear reads them)
        Buf proj(this, (size_t)rows * 2 * hidden * 4);
        gemm(x, (int)rows, bt, bias, cin, 2 * hidden, proj.f(), 2 * hidden, nullptr, 0, dt, 0, x3, nullptr);
        {
            ProfScope ps(this, PC_GEGLU, 0, (double)rows * hidden * (8.0 + (out3 ? 6.0 : 4.0)));
            if (out3) SDMI_HIP(launch_geglu_planes(proj.f(), out3, rows, hidden, stream_));
            else SDMI_HIP(launch_geglu(proj.f(), out, rows, hidden, stream_));
        }
        count_kernel();
        return;
    }
    const size_t es = dt ? 2 : 4;
    // the fused form needs a large-tile kernel with an even fragment count per wave (256x256 or 256x128 tiles, no split-K)
    int cfg = -1;
    const bool eligible = opt_geglu_fuse_ && hidden % 8 == 0 && cin % (dt ? 64 : 32) == 0 && (dt ? opt_gemm_bf16x_ : opt_gemm_x32_);
    if (eligible) {
        const long long mt = (rows + 255) / 256;
        const long long t256 = mt * ((hidden + 127) / 128), t128 = mt * ((hidden + 63) / 64);   // tiles with 256x256 / 256x128
        // measured (same box, --opt geglu_fuse=0/1): at batch 1 the 256-wide tiles quantise badly against 256 CUs (320 tiles =
        // two rounds) and the fused form LOSES 2.6 % end to end in fp32; with >= 4 rounds it wins ~1 % (bf16, batch 8)
        if (t256 >= 1024 || opt_geglu_fuse_ == 3) cfg = 101;
        else if (t128 >= 1024 || opt_geglu_fuse_ == 2) cfg = 102;
        // precision = 0 with the split kernels: their even-fragment tiles (128x256s / 256x128s / 128x128s = 128 / 64 / 64 output
        // columns per tile); geglu_fuse = 4 / 5 / 6 force them
        if (!dt && opt_gemm_f32s_ && split_planes(bt)) {
            if (opt_geglu_fuse_ == 4) cfg = 203;
            else if (opt_geglu_fuse_ == 5) cfg = 202;
            else if (opt_geglu_fuse_ == 6) cfg = 205;      // (measured at batch 1: 3.56 / 3.58 / 3.54 img/s against 3.66 unfused)
        }
    }
    if (cfg >= 0) {
        ConvGemm p{};
        p.A = x; p.Bt = bt; p.C = out; p.bias = bias;
        p.M = (int)rows; p.N = hidden; p.K = cin;
        p.NB = 1; p.Hs = 1; p.Ws = (int)rows; p.Cin = cin; p.Ho = 1; p.Wo = (int)rows;
        p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.ups = 0;
        p.ldc = hidden; p.ldr = hidden; p.a_ld = cin; p.b_ld = cin; p.rowvec_stride = 0; p.CS = 32;
        p.out_mode = 0;
        p.geglu = 1;
        launch_gemm(p, dt, cfg, 1);
        return;
    }
    Buf proj(this, (size_t)rows * 2 * hidden * es);
    gemm(x, (int)rows, bt, bias, cin, 2 * hidden, proj.f(), 2 * hidden, nullptr, 0, dt);
    {
        ProfScope ps(this, PC_GEGLU, 0, (double)rows * hidden * (dt ? 6.0 : 12.0));
        if (dt) SDMI_HIP(launch_geglu_bf16(proj.p, out, rows, hidden, stream_));
        else SDMI_HIP(launch_geglu(proj.f(), out, rows, hidden, stream_));
    }
    count_kernel();
}

void Engine::group_norm(const NormW& w, const Act& x, Act& y, bool silu) {
    const int hw = x.h * x.w;
    if (x.dt != y.dt) throw Error(SDMI_ERR_STATE, "group_norm: in/out storage types disagree");
    Buf part(this, x.dt ? gn_partials_bytes_bf16(x.n, hw, x.c, gn_tune_) : gn_partials_bytes(x.n, hw, x.c, opt_gn32_min_wgs_));
    ProfScope ps(this, PC_GROUP_NORM, 0, 2.0 * (double)x.bytes(), 2);  // algorithmic: one read + one write; two launches (statistics, apply)
    ps.set_tag("group_norm n%d hw%d c%d%s", x.n, hw, x.c, silu ? " silu" : "");
    if (y.view) throw Error(SDMI_ERR_STATE, "group_norm: output must be dense");
    if (!x.p) throw Error(SDMI_ERR_STATE, "group_norm: the input must exist as fp32");
    if (y.p3 && !y.p) {   // the consumer is a plane GEMM: the normalised tensor is written as three bf16 planes only
        if (x.dt) throw Error(SDMI_ERR_STATE, "group_norm: planes are an fp32-engine format");
        SDMI_HIP(launch_group_norm_planes(x.p, y.p3, w.gamma, w.beta, x.n, hw, x.c, x.stride(), 32, w.eps, silu, part.p, stream_, opt_gn32_min_wgs_));
    }
    else if (x.dt) SDMI_HIP(launch_group_norm_bf16(x.p, y.p, w.gamma, w.beta, x.n, hw, x.c, x.stride(), 32, w.eps, silu, part.p, stream_, gn_tune_));
    else SDMI_HIP(launch_group_norm(x.p, y.p, w.gamma, w.beta, x.n, hw, x.c, x.stride(), 32, w.eps, silu, part.p, stream_, opt_gn32_min_wgs_));
    count_kernel(); count_kernel();
}

void Engine::layer_norm(const NormW& w, const float* x, long long rows, float* y, int dt, void* y3) {
    if (dt < 0) dt = edt();
    ProfScope ps(this, PC_LAYER_NORM, 0, 2.0 * (double)rows * w.c * (dt ? 2.0 : 4.0));
    ps.set_tag("layer_norm rows%lld c%d", rows, w.c);
    if (y3) {
        if (dt) throw Error(SDMI_ERR_STATE, "layer_norm: planes are an fp32-engine format");
        SDMI_HIP(launch_layer_norm_planes(x, y3, w.gamma, w.beta, (int)rows, w.c, w.eps, stream_));
    } else if (dt) SDMI_HIP(launch_layer_norm_bf16(x, y, w.gamma, w.beta, (int)rows, w.c, w.eps, stream_));
    else SDMI_HIP(launch_layer_norm(x, y, w.gamma, w.beta, (int)rows, w.c, w.eps, stream_));
    count_kernel();
}

// qkv_attention (attention.rs:5-45).  Head dims with a fused instance use the flash
// kernel; others (the VAE's single 512-wide head) run QK^T -> row softmax -> PV on
// the GEMM kernel, one (batch, head) at a time.
void Engine::attention(const float* q, int ldq, long long q_bs, const float* k, int ldk, long long k_bs,
                       const float* v, int ldv, long long v_bs, float* o, int ldo, long long o_bs, int n, int nq,
                       int nk, int n_head, int d_head, const int* kv_len_dev, const int* kv_len_host,
                       const float* mask, int mask_ld, int dt, void* o3, bool q_log2) {
    if (dt < 0) dt = edt();
    if (nq <= 0 || nk <= 0) throw Error(SDMI_ERR_INVALID, "attention: empty sequence");
    if (o3 && (dt || !attn_supported_head_dim(d_head) || (n_head * d_head) % 32)) throw Error(SDMI_ERR_STATE, "attention: plane output needs a fused fp32 kernel");
    // q_log2 (stated by the caller, Engine::q_prescaled): q arrives in log2 units (attn_bf16_q_scale: folded into the query weights at load, applied by
    // qkv_attention_dev's conversion); the bf16 kernel needs no scale, the widened fp32 kernel (attn_bf16=0) gets scale^2 = ln 2
    if (q_log2 != q_prescaled(dt, d_head)) throw Error(SDMI_ERR_STATE, "attention: the caller's statement about the query's scale does not match the storage type / head dim");
    const float scale = q_log2 ? 0.83255461115769775635f : (float)std::pow((double)d_head, -0.25);
    if (attn_supported_head_dim(d_head)) {
        AttnParams p{};
        p.q = q; p.k = k; p.v = v; p.o = o; p.kv_len = kv_len_dev; p.mask = mask; p.mask_ld = mask_ld;
        p.n = n; p.n_head = n_head; p.nq = nq; p.nk = nk; p.d_head = d_head;
        p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
        p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs; p.scale = scale;
        p.bf16 = dt;
        p.q_log2 = q_log2 ? 1 : 0;
        p.o3 = o3; p.ldo3 = (n_head * d_head / 32) * 192;
        p.pack_tail = opt_attn_pack_tail_;
        p.variant = opt_attn_bf16_variant_;
        if (dt && mask) throw Error(SDMI_ERR_UNSUPPORTED, "attention: additive mask is fp32-only");
        const double fl = 4.0 * n * n_head * (double)nq * nk * d_head;
        const bool on_split = !dt && opt_attn_split_ && attn_split_supported(p);
        // Key slices (round 5; fp32, no mask): at batch 1 the 32 x 32 level's self attention is 128 workgroups and the 16 x 16 level's 64 -- most CUs idle while each
        // workgroup walks every key.  S slices of the keys run as S x the workgroups (blockIdx.z) and a merge launch combines them in slice order.  S = what fills
        // 256 CUs, at least two K / V tiles per slice, at most 8; option attn_kv_splits: 0 = this rule, 1 = never, S = forced.
        int kv_splits = 1;
        if (!dt && !mask && opt_attn_kv_splits_ != 1) {
            const long long wgs = (long long)((nq + (on_split ? 127 : 63)) / (on_split ? 128 : 64)) * n * n_head;   // the 4-wave workgroups these sizes get
            const int tiles = (nk + attn_f32_kv_tile(p) - 1) / attn_f32_kv_tile(p);
            long long s_auto = wgs > 0 && wgs <= 128 ? 256 / wgs : 1;
            s_auto = std::min<long long>(std::min<long long>(s_auto, tiles / 2), 8);
            if (on_split && opt_attn_kv_prefer8_) {   // k_attn_split.hip: enough slices that the 8-wave form fills the chip beat fewer slices of the 4-wave form
                const long long wgs8 = (long long)((nq + 255) / 256) * n * n_head;
                const long long s8 = std::min<long long>(std::min<long long>(wgs8 < 256 ? (256 + wgs8 - 1) / wgs8 : 1, tiles / 2), 8);
                if (s8 >= 2 && wgs8 * s8 >= 256) s_auto = s8;
            }
            kv_splits = opt_attn_kv_splits_ > 1 ? std::min(opt_attn_kv_splits_, std::max(1, tiles)) : (int)std::max<long long>(1, s_auto);
        }
        std::unique_ptr<Buf> part_o, part_ml;
        if (kv_splits > 1) {
            part_o.reset(new Buf(this, (size_t)kv_splits * n * nq * n_head * d_head * sizeof(float)));
            part_ml.reset(new Buf(this, (size_t)kv_splits * n * n_head * nq * 2 * sizeof(float)));
            p.kv_splits = kv_splits; p.part_o = part_o->f(); p.part_ml = part_ml->f();
        }
        ProfScope ps(this, PC_ATTENTION, fl, 0, kv_splits > 1 ? 2 : 1);
        ps.set_tag("attention n%d nq%d nk%d h%d d%d slices=%d", n, nq, nk, n_head, d_head, kv_splits);
        if (dt && opt_attn_bf16_ && (d_head == 40 || d_head == 80 || d_head == 160)) SDMI_HIP(launch_attention_bf16(p, stream_));
        else if (on_split) SDMI_HIP(launch_attention_split(p, stream_));
        else SDMI_HIP(launch_attention(p, stream_));
        count_kernel(fl);
        if (kv_splits > 1) {
            SDMI_HIP(launch_attention_combine(p, stream_));
            count_kernel();
        }
        return;
    }
    if (mask) throw Error(SDMI_ERR_UNSUPPORTED, "attention: additive mask is only supported for head dims 40/80/160");
    if (d_head % 32) throw Error(SDMI_ERR_UNSUPPORTED, "attention: head dim must be 40/80/160 or a multiple of 32");
    if (dt && (d_head % 64)) throw Error(SDMI_ERR_UNSUPPORTED, "bf16 attention (unfused path): head dim must be a multiple of 64");
    for (int b = 0; b < n; ++b) {
        const int nkb = kv_len_host ? kv_len_host[b] : nk;
        if (nkb % (dt ? 64 : 32)) throw Error(SDMI_ERR_UNSUPPORTED, "attention (unfused path): key count must be a multiple of 32 (64 for bf16)");
        Buf s(this, (size_t)nq * nkb * sizeof(float));            // scores stay fp32 in both precisions
        Buf pb(this, dt ? (size_t)nq * nkb * 2 : 256);             // bf16 probabilities (precision = 1)
        Buf vt(this, (size_t)d_head * nkb * (dt ? 2 : 4));
        for (int h = 0; h < n_head; ++h) {
            const float* qb = adv(q, b * q_bs + h * d_head, dt);
            const float* kb = adv(k, b * k_bs + h * d_head, dt);
            const float* vb = adv(v, b * v_bs + h * d_head, dt);
            float* ob = adv(o, b * o_bs + h * d_head, dt);
            ConvGemm g{};
            g.A = qb; g.Bt = kb; g.C = s.f();
            g.M = nq; g.N = nkb; g.K = d_head; g.NB = 1; g.Hs = 1; g.Ws = nq; g.Cin = d_head; g.Ho = 1; g.Wo = nq;
            g.KH = g.KW = 1; g.stride = 1; g.ldc = nkb; g.ldr = nkb; g.a_ld = ldq; g.b_ld = ldk; g.CS = 32;
            g.out_mode = dt ? 1 : 0;
            launch_gemm(g, dt);
            if (dt) { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_softmax_rows_f32_to_bf16(s.f(), pb.p, nq, nkb, scale * scale, stream_)); }
            else { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_softmax_rows(s.f(), nq, nkb, scale * scale, stream_)); }
            count_kernel();
            if (dt) { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_transpose2d_bf16(vb, vt.p, nkb, d_head, ldv, stream_)); }
            else { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_transpose2d(vb, vt.f(), nkb, d_head, ldv, stream_)); }
            count_kernel();
            ConvGemm g2{};
            g2.A = dt ? pb.f() : s.f(); g2.Bt = vt.f(); g2.C = ob;
            g2.M = nq; g2.N = d_head; g2.K = nkb; g2.NB = 1; g2.Hs = 1; g2.Ws = nq; g2.Cin = nkb; g2.Ho = 1; g2.Wo = nq;
            g2.KH = g2.KW = 1; g2.stride = 1; g2.ldc = ldo; g2.ldr = ldo; g2.a_ld = nkb; g2.b_ld = nkb; g2.CS = 32;
            launch_gemm(g2, dt);
        }
    }
}

// =============================================================================
// composite blocks
// =============================================================================
// ResBlock::forward (unet/mod.rs:713-733) / ResnetBlock::forward (autoencoder/mod.rs:514-527)
void Engine::res_block(const ResW& w, const Act& x, Act& y, int step) {
    Range rng(this, "ResBlock");
    Act h2 = new_act(x.n, x.h, x.w, w.cout);
    const float* rowvec = nullptr;
    if (w.has_embed) rowvec = us_.temb.at(w.temb_index) + (size_t)step * w.cout;  // shared by the batch (one timestep)
    if (use_fp8(w.conv_in, x)) {   // precision = 2: the normalisation writes MXFP8, the conv runs on the MX-scaled matrix instruction
        ActQ q1 = new_actq(x.n, x.h, x.w, x.c);
        group_norm_fp8(w.norm_in, x, q1, true);
        conv_fp8(w.conv_in, q1, h2, rowvec, nullptr);
        release(q1);
    } else {
        // the normalised tensors have one consumer, a 3x3 convolution: on the fp32 engine they exist only as bf16 planes (k_gemm3p.hip)
        Act h1 = plane_gemm(x.c, w.cout) ? new_act3(x.n, x.h, x.w, x.c, 2) : new_act(x.n, x.h, x.w, x.c);
        group_norm(w.norm_in, x, h1, true);
        conv(w.conv_in, h1, h2, 1, 0, rowvec, 0, nullptr);
        release(h1);
    }
    // the shortcut's result is the residual of conv_out: it goes through y's fp32 buffer, or through a temporary when y exists as planes only
    Act sk{};
    if (w.has_skip) {
        if (y.p) { sk = y; sk.p3 = nullptr; sk.view = true; }
        else sk = new_act(y.n, y.h, y.w, y.c);
        if (use_fp8_wide(w.skip.bt8, x.rows()) && x.dt == 1 && x.c % 32 == 0) {
            ActQ xq = new_actq(x.n, x.h, x.w, x.c);
            quantize(x, xq);
            conv_fp8(w.skip, xq, sk, nullptr, nullptr);
            release(xq);
        } else {
            conv(w.skip, x, sk, 1, 0, nullptr, 0, nullptr);
        }
    }
    const Act* resid = w.has_skip ? &sk : &x;
    if (use_fp8(w.conv_out, h2)) {
        ActQ q3 = new_actq(x.n, x.h, x.w, w.cout);
        group_norm_fp8(w.norm_out, h2, q3, true);
        release(h2);
        conv_fp8(w.conv_out, q3, y, nullptr, resid);
        release(q3);
    } else {
        Act h3 = plane_gemm(w.cout, w.cout) ? new_act3(x.n, x.h, x.w, w.cout, 2) : new_act(x.n, x.h, x.w, w.cout);
        group_norm(w.norm_out, h2, h3, true);
        release(h2);
        conv(w.conv_out, h3, y, 1, 0, nullptr, 0, resid);
        release(h3);
    }
    if (w.has_skip) release(sk);
}

// ---- precision = 2: MXFP8 GroupNorm output + 3x3 convolution (k_fp8.hip) ------------------------------------------------
bool Engine::use_fp8(const ConvW& w, const Act& x) const {
    return fp8_ && opt_fp8_convs_ && w.bt8 && x.dt == 1 && x.rows() >= opt_fp8_min_rows_;
}

ActQ Engine::new_actq(int n, int h, int w, int c) {
    ActQ a; a.n = n; a.h = h; a.w = w; a.c = c; a.cp = (c + 127) / 128 * 128;
    a.q = pool_.alloc((size_t)a.rows() * a.cp);
    a.s = pool_.alloc((size_t)a.rows() * (a.cp / 32));
    return a;
}
void Engine::release(ActQ& a) {
    if (a.q) pool_.free(a.q);
    if (a.s) pool_.free(a.s);
    a.q = a.s = nullptr;
}

void Engine::group_norm_fp8(const NormW& w, const Act& x, ActQ& y, bool silu) {
    const int hw = x.h * x.w;
    if (x.dt != 1 || y.c != x.c) throw Error(SDMI_ERR_STATE, "group_norm_fp8: bf16 input of matching width expected");
    Buf part(this, gn_partials_bytes_bf16(x.n, hw, x.c, gn_tune_));
    ProfScope ps(this, PC_GROUP_NORM, 0, (double)x.bytes() + (double)x.rows() * (y.cp + y.cp / 32), 2);
    SDMI_HIP(launch_group_norm_fp8(x.p, y.q, y.s, w.gamma, w.beta, x.n, hw, x.c, x.stride(), 32, w.eps, silu, part.p, stream_, &gn_tune_));
    count_kernel(); count_kernel();
}

void Engine::conv_fp8(const ConvW& w, const ActQ& x, Act& y, const float* rowvec, const Act* resid, int stride, int ups) {
    if (x.c != w.cin || (w.k != 3 && w.k != 1) || !w.bt8) throw Error(SDMI_ERR_STATE, "conv_fp8: not an MXFP8-packed convolution");
    const int pad = w.k == 3 ? 1 : 0;
    const int hin = x.h << ups, win = x.w << ups;
    const int ho = (hin + 2 * pad - w.k) / stride + 1, wo = (win + 2 * pad - w.k) / stride + 1;
    if (y.n != x.n || y.h != ho || y.w != wo || y.c != w.cout || y.dt != 1) throw Error(SDMI_ERR_INVALID, "conv_fp8: output shape / type mismatch");
    if (resid && (resid->rows() != y.rows() || resid->c != y.c || resid->dt != 1)) throw Error(SDMI_ERR_STATE, "conv_fp8: residual shape / type mismatch");
    ConvGemm p{};
    p.A = reinterpret_cast<const float*>(x.q); p.Bt = w.bt8; p.C = y.p; p.bias = w.bias; p.rowvec = rowvec; p.resid = resid ? resid->p : nullptr;
    p.a_scale = x.s; p.b_scale = w.bs8;
    p.M = x.n * ho * wo; p.N = w.cout; p.K = x.cp * w.k * w.k;
    p.NB = x.n; p.Hs = x.h; p.Ws = x.w; p.Cin = x.cp; p.Ho = ho; p.Wo = wo;
    p.KH = w.k; p.KW = w.k; p.stride = stride; p.pad = pad; p.ups = ups;
    p.ldc = y.stride(); p.ldr = resid ? resid->stride() : y.stride(); p.a_ld = x.cp; p.b_ld = p.K; p.rowvec_stride = 0; p.CS = 128;
    launch_fp8(p, 2.0 * p.M * (double)p.N * w.cin * w.k * w.k);   // algorithmic (unpadded) work
}

// C[rows][n_rows_w] (bf16) = x W^T + bias (+ resid): a Linear layer on MXFP8 operands (n_rows_w = w.cout, or 3 cout for the packed q | k | v)
void Engine::gemm_fp8(const ActQ& x, const LinW& w, int n_rows_w, void* C, int ldc, const float* resid, int ldr) {
    if (x.c != w.cin || !w.bt8) throw Error(SDMI_ERR_STATE, "gemm_fp8: not an MXFP8-packed Linear layer");
    ConvGemm p{};
    p.A = reinterpret_cast<const float*>(x.q); p.Bt = w.bt8; p.C = reinterpret_cast<float*>(C); p.bias = w.bias; p.resid = resid;
    p.a_scale = x.s; p.b_scale = w.bs8;
    p.M = (int)x.rows(); p.N = n_rows_w; p.K = x.cp;
    p.NB = 1; p.Hs = 1; p.Ws = p.M; p.Cin = x.cp; p.Ho = 1; p.Wo = p.M;
    p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.ups = 0;
    p.ldc = ldc; p.ldr = ldr; p.a_ld = x.cp; p.b_ld = p.K; p.rowvec_stride = 0; p.CS = 128;
    launch_fp8(p, 2.0 * p.M * (double)p.N * w.cin);
}

void Engine::launch_fp8(ConvGemm& p, double flops) {
    p.out_mode = 0;
    p.variant = opt_gemm_bf16x_variant_;
    p.zero_page = zero_page_;
    p.kt_total = p.K / 128;
    if ((unsigned long long)p.NB * p.Hs * p.Ws * (unsigned long long)p.a_ld >= 0xFFFFFFE0ull || (unsigned long long)p.N * p.K >= 0xFFFFFFE0ull) throw Error(SDMI_ERR_UNSUPPORTED, "fp8 GEMM: operand larger than 4 GiB");
    // tile + split-K: rounds of workgroups on 256 CUs x the time of one tile at the rate each tile shape sustains when the
    // chip is full (tools/bench_gemm_fp8.py on MI355X: 256x320 2.4, 256x256 2.1, 256x128 1.7 PFLOP/s); K is split only when
    // the tiles would leave half of the chip or more idle
    static const double kRate[kNumGemmTilesQ] = {2400.0, 2100.0, 1700.0};
    int cfg = opt_fp8_tile_, splits = 1;
    auto plan = [&](int c, int* sp) {
        const int bm = gemm_tile_info_q(c).bm, bn = gemm_tile_info_q(c).bn;
        const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
        int s_ = 1;
        // (round 6: <= 128 -- half a round of tiles is split too: M = 8192, N = 1280 on 256 x 320 tiles is 128 workgroups; K = 11520: 167.5 -> 143.0 us, K = 23040: 314 -> 240, profiles/r06q_fp8_shapes.txt)
        if (tiles <= 128) s_ = (int)std::max<long long>(1, std::min<long long>(p.kt_total / 4, (256 + tiles - 1) / tiles));
        *sp = s_;
        const double rounds = (double)((tiles * s_ + 255) / 256);
        return rounds * (double)bm * bn / kRate[c] / s_ + (s_ > 1 ? 0.15 * (double)bm * bn / kRate[c] : 0.0);
    };
    if (cfg < 0) {
        double best = 1e300;
        for (int c = 0; c < kNumGemmTilesQ; ++c) {
            int sp;
            const double t = plan(c, &sp);
            if (t < best) { best = t; cfg = c; splits = sp; }
        }
    } else {
        if (cfg >= kNumGemmTilesQ) throw Error(SDMI_ERR_INVALID, "fp8_tile out of range");
        (void)plan(cfg, &splits);
    }
    if (opt_force_splits_ > 0) splits = opt_force_splits_;
    splits = std::max(1, std::min(splits, p.kt_total));
    p.kt_per_split = (p.kt_total + splits - 1) / splits;
    splits = (p.kt_total + p.kt_per_split - 1) / p.kt_per_split;
    p.splits = splits;
    p.resid_acc = 0;   // (as Engine::launch_gemm)
    if (splits == 1 && (p.N % 8) == 0 && (p.ldc % 8) == 0) {
        if ((opt_resid_acc_ & 1) && p.resid && (p.ldr % 4) == 0) p.resid_acc |= 1;
        if ((opt_resid_acc_ & 2) && (p.bias || p.rowvec) && (p.rowvec_stride % 4) == 0) p.resid_acc |= 2;
    }
    if (record_shapes_) {   // option dump_choices: the MXFP8 launches are listed with their own tag, so a test can pin WHICH layers run on fp8 operands
        char ck[128];
        std::snprintf(ck, sizeof ck, "%d,%d,%d k%d s%d u%d W%d fp8 cfg=%d splits=%d", p.M, p.N, p.K, p.KH, p.stride, p.ups, p.Ws, cfg, splits);
        ++choice_counts_[ck];
    }
    // algorithmic bytes: e4m3 operands + one E8M0 scale byte per 32 elements, bf16 result
    const double fp8_bytes = ((double)p.NB * p.Hs * p.Ws * p.a_ld + (double)p.N * p.K) * (1.0 + 1.0 / 32.0) + (double)p.M * p.N * 2.0;
    if (splits == 1) {
        ProfScope ps(this, PC_CONV_FP8, flops, fp8_bytes);
        ps.set_tag("gemm_fp8 %d,%d,%d k%d%s%s cfg=%d splits=1", p.M, p.N, p.K, p.KH, p.resid ? " resid" : "", p.rowvec ? " rowvec" : "", cfg);
        SDMI_HIP(launch_conv_gemm_fp8x(p, cfg, stream_));
        count_kernel(flops);
    } else {
        p.slab_stride = (long long)p.M * p.N;
        Buf slab(this, (size_t)splits * p.slab_stride * sizeof(float));
        p.slabs = slab.f();
        {
            ProfScope ps(this, PC_CONV_FP8, flops, fp8_bytes);
            ps.set_tag("gemm_fp8 %d,%d,%d k%d cfg=%d splits=%d", p.M, p.N, p.K, p.KH, cfg, splits);
            SDMI_HIP(launch_conv_gemm_fp8x(p, cfg, stream_));
        }
        count_kernel(flops);
        ProfScope ps(this, PC_SPLITK_REDUCE, 0, (double)(splits + 1) * p.slab_stride * 4.0);
        SDMI_HIP(launch_splitk_reduce_bf16(p, stream_));
        count_kernel();
    }
}

// a convolution whose input is a raw (not normalised) activation -- the down / up convolutions (unet/mod.rs:397,425; autoencoder/mod.rs:319):
// precision = 2 with option fp8_linear quantises the input in front of it, everything else is conv()
void Engine::conv_raw(const ConvW& w, const Act& x, Act& y, int stride, int ups) {
    if (use_fp8_wide(w.bt8, y.rows()) && x.dt == 1 && y.dt == 1 && x.p && x.c % 32 == 0) {
        ActQ xq = new_actq(x.n, x.h, x.w, x.c);
        quantize(x, xq);
        conv_fp8(w, xq, y, nullptr, nullptr, stride, ups);
        release(xq);
        return;
    }
    conv(w, x, y, stride, ups, nullptr, 0, nullptr);
}

void Engine::quantize(const Act& x, ActQ& y) {
    if (x.dt != 1 || !x.p || y.c != x.c || y.rows() != x.rows()) throw Error(SDMI_ERR_STATE, "quantize: bf16 input of matching shape expected");
    ProfScope ps(this, PC_OTHER, 0, (double)x.rows() * (x.c * 2.0 + y.cp * 1.03));
    SDMI_HIP(launch_quantize_bf16_fp8(x.p, y.q, y.s, x.rows(), x.c, x.stride(), stream_));
    count_kernel();
}

void Engine::layer_norm_fp8(const NormW& w, const float* x, long long rows, ActQ& y) {
    if (y.c != w.c || y.rows() != rows) throw Error(SDMI_ERR_STATE, "layer_norm_fp8: output shape mismatch");
    ProfScope ps(this, PC_LAYER_NORM, 0, (double)rows * (w.c * 2.0 + y.cp * 1.03));
    SDMI_HIP(launch_layer_norm_fp8(x, y.q, y.s, w.gamma, w.beta, (int)rows, w.c, w.eps, stream_));
    count_kernel();
}

// SpatialTransformer::forward (unet/mod.rs:462-480) + TransformerBlock (:522-526) +
// MultiHeadAttention (:642-652) + MLP/GEGLU (:552-591).  NHWC makes the reference's
// two NCHW<->token transposes disappear.
void Engine::spatial_transformer(const SpatialW& w, const Act& x, Act& y) {
    Range rng(this, "SpatialTransformer");
    // y.n == 2 x.n: the shared prefix of a CFG pair (unet_run): x holds ONE copy of the two halves' identical input; everything in front of the cross attention --
    // the first place the text context enters -- is computed once and duplicated there
    if (y.n != x.n && y.n != 2 * x.n) throw Error(SDMI_ERR_STATE, "spatial_transformer: batch mismatch");
    auto duplicate = [&](const Act& src) {   // [n] -> [2n] (dense, same type): both halves = src
        Act d2 = new_act(2 * src.n, src.h, src.w, src.c);
        ProfScope ps_o(this, PC_OTHER, 0, 3.0 * (double)src.bytes());
        SDMI_HIP(launch_repeat_rows(src.p, d2.p, 2, (long long)(src.bytes() / 4), stream_));
        count_kernel();
        return d2;
    };
    const int C = w.c, hw = x.h * x.w;
    const int heads = cfg_.n_head, d = C / heads;
    if (y.n != x.n && use_fp8_wide(w.proj_in.bt8, y.rows()) && x.dt == 1 && y.dt == 1) {   // option fp8_linear: no shared form -- duplicate first
        Act x2 = duplicate(x);
        spatial_transformer(w, x2, y);
        release(x2);
        return;
    }
    const int nb = y.n, n1 = x.n;            // n1: samples of the part in front of the cross attention
    const bool share = nb != n1;
    const long long M = (long long)nb * hw, M1 = (long long)n1 * hw;
    if (use_fp8_wide(w.proj_in.bt8, M) && w.attn1.q.bt8 && w.attn1.out.bt8 && w.attn2.q.bt8 && w.attn2.out.bt8 && w.geglu_proj.bt8 && w.mlp_lin.bt8 &&
        w.proj_out.bt8 && x.dt == 1 && y.dt == 1) {
        // precision = 2, option fp8_linear: every GEMM of the block on MXFP8 operands.  Their inputs are quantised by the kernel that
        // produces them (GroupNorm, LayerNorm, the GEGLU gate) or by quantize() (attention outputs, the hidden state in front of proj_out);
        // the residual stream h, q / k / v and the attention itself stay bf16 (why attention is not fp8).
        ActQ gq = new_actq(x.n, x.h, x.w, C);
        group_norm_fp8(w.norm, x, gq, false);
        Act h = new_act(x.n, x.h, x.w, C);
        conv_fp8(w.proj_in, gq, h, nullptr, nullptr);
        release(gq);
        {
            ActQ lnq = new_rowsq(M, C), aq = new_rowsq(M, C);
            Buf q(this, (size_t)M * C * 2), a(this, (size_t)M * C * 2);
            Act av; av.p = a.f(); av.n = 1; av.h = 1; av.w = (int)M; av.c = C; av.dt = 1;
            layer_norm_fp8(w.ln1, h.p, M, lnq);
            {
                Buf qkv(this, (size_t)M * 3 * C * 2);
                gemm_fp8(lnq, w.attn1.q, 3 * C, qkv.p, 3 * C, nullptr, 0);      // q | k | v: the packed [3C][Kp] weight
                const long long bs3 = (long long)hw * 3 * C;
                attention(qkv.f(), 3 * C, bs3, adv(qkv.f(), C, 1), 3 * C, bs3, adv(qkv.f(), 2 * C, 1), 3 * C, bs3, a.f(), C,
                          (long long)hw * C, nb, hw, hw, heads, d, nullptr, nullptr, nullptr, 0, -1, nullptr, q_prescaled(edt(), d));
            }
            quantize(av, aq);
            gemm_fp8(aq, w.attn1.out, C, h.p, C, h.p, C);
            layer_norm_fp8(w.ln2, h.p, M, lnq);
            gemm_fp8(lnq, w.attn2.q, C, q.p, C, nullptr, 0);
            const long long cbs = (long long)us_.t_max * C;
            attention(q.f(), C, (long long)hw * C, us_.kc.at(w.ctx_index), C, cbs, us_.vc.at(w.ctx_index), C, cbs, a.f(),
                      C, (long long)hw * C, nb, hw, us_.t_max, heads, d, us_.kv_len_dev, us_.kv_len_host.data(), nullptr, 0, -1, nullptr, q_prescaled(edt(), d));
            quantize(av, aq);
            gemm_fp8(aq, w.attn2.out, C, h.p, C, h.p, C);
            layer_norm_fp8(w.ln3, h.p, M, lnq);
            {
                Buf proj(this, (size_t)M * 8 * C * 2);
                gemm_fp8(lnq, w.geglu_proj, 8 * C, proj.p, 8 * C, nullptr, 0);
                ActQ uq = new_rowsq(M, 4 * C);
                {
                    ProfScope ps(this, PC_GEGLU, 0, (double)M * (8.0 * C * 2.0 + 4.0 * C * 1.03));
                    SDMI_HIP(launch_geglu_fp8(proj.p, uq.q, uq.s, M, 4 * C, stream_));
                    count_kernel();
                }
                gemm_fp8(uq, w.mlp_lin, C, h.p, C, h.p, C);
                release(uq);
            }
            release(lnq); release(aq);
        }
        ActQ hq = new_actq(x.n, x.h, x.w, C);
        quantize(h, hq);
        release(h);
        conv_fp8(w.proj_out, hq, y, nullptr, &x);
        release(hq);
        return;
    }
    // fp32 engine: every tensor whose only consumer is a GEMM (the normalised activations, the attention outputs, the gated MLP
    // hidden state, the block's last hidden state) is written by its producer as three bf16 planes -- what k_gemm3p.hip reads
    const bool pl = plane_gemm(C, C) && attn_supported_head_dim(d);
    Act g = pl ? new_act3(x.n, x.h, x.w, C, 2) : new_act(x.n, x.h, x.w, C);
    group_norm(w.norm, x, g, false);
    Act h = new_act(x.n, x.h, x.w, C);
    conv(w.proj_in, g, h, 1, 0, nullptr, 0, nullptr);
    release(g);
    Act hp = pl ? new_act3(nb, x.h, x.w, C, 2) : Act{};      // the hidden state after the MLP: read by proj_out only
    Act x2{};                                                 // shared prefix: the block input once per half (proj_out's residual)
    {
        const size_t es = esz();
        const size_t row3 = (size_t)(C / 32) * 192;
        Buf ln(this, pl ? (size_t)M * row3 : (size_t)M * C * es), q(this, (size_t)M * C * es), a(this, pl ? (size_t)M * row3 : (size_t)M * C * es);
        float* lnf = pl ? nullptr : ln.f();
        void* ln3 = pl ? ln.p : nullptr;
        float* af = pl ? nullptr : a.f();
        void* a3 = pl ? a.p : nullptr;
        // self attention: q, k, v in one GEMM (N = 3C) on the packed [3C][C] weight
        layer_norm(w.ln1, h.p, M1, lnf, -1, ln3);
        {
            Buf qkv(this, (size_t)M1 * 3 * C * es);
            gemm(lnf, (int)M1, w.attn1.q.bt, nullptr, C, 3 * C, qkv.f(), 3 * C, nullptr, 0, -1, 0, ln3);
            const long long bs3 = (long long)hw * 3 * C;
            attention(qkv.f(), 3 * C, bs3, adv(qkv.f(), C, edt()), 3 * C, bs3, adv(qkv.f(), 2 * C, edt()), 3 * C, bs3, af, C,
                      (long long)hw * C, n1, hw, hw, heads, d, nullptr, nullptr, nullptr, 0, -1, a3, q_prescaled(edt(), d));
        }
        gemm(af, (int)M1, w.attn1.out.bt, w.attn1.out.bias, C, C, h.p, C, h.p, C, -1, 0, a3);
        if (share) {   // from here on the halves differ: the hidden state and the block input, once per half
            Act h2 = duplicate(h);
            release(h);
            h = h2;
            x2 = duplicate(x);
        }
        // cross attention against the hoisted K/V of the text context
        layer_norm(w.ln2, h.p, M, lnf, -1, ln3);
        gemm(lnf, (int)M, w.attn2.q.bt, nullptr, C, C, q.f(), C, nullptr, 0, -1, 0, ln3);
        const long long cbs = (long long)us_.t_max * C;
        attention(q.f(), C, (long long)hw * C, us_.kc.at(w.ctx_index), C, cbs, us_.vc.at(w.ctx_index), C, cbs, af,
                  C, (long long)hw * C, nb, hw, us_.t_max, heads, d, us_.kv_len_dev, us_.kv_len_host.data(), nullptr, 0, -1, a3, q_prescaled(edt(), d));
        gemm(af, (int)M, w.attn2.out.bt, w.attn2.out.bias, C, C, h.p, C, h.p, C, -1, 0, a3);
        // GEGLU MLP
        layer_norm(w.ln3, h.p, M, lnf, -1, ln3);
        {
            Buf u(this, pl ? (size_t)M * 4 * row3 : (size_t)M * 4 * C * es);
            gemm_geglu(lnf, M, w.geglu_proj.bt, w.geglu_proj.bias, C, 4 * C, pl ? nullptr : u.f(), -1, ln3, pl ? u.p : nullptr);
            if (pl) gemm(nullptr, (int)M, w.mlp_lin.bt, w.mlp_lin.bias, 4 * C, C, nullptr, C, h.p, C, -1, 0, u.p, hp.p3);   // h + mlp -> planes only
            else gemm(u.f(), (int)M, w.mlp_lin.bt, w.mlp_lin.bias, 4 * C, C, h.p, C, h.p, C);
        }
    }
    const Act& xr = share ? x2 : x;
    if (pl) { conv(w.proj_out, hp, y, 1, 0, nullptr, 0, &xr); release(hp); }
    else conv(w.proj_out, h, y, 1, 0, nullptr, 0, &xr);
    release(h);
    if (share) release(x2);
}

// ConvSelfAttentionBlock::forward (autoencoder/mod.rs:563-607)
void Engine::vae_attn(const VaeAttnW& w, const Act& x, Act& y) {
    Range rng(this, "ConvSelfAttentionBlock");
    const int C = w.c, hw = x.h * x.w;
    Act g = new_act(x.n, x.h, x.w, C);
    group_norm(w.norm, x, g, false);
    Act q = new_act(x.n, x.h, x.w, C), k = new_act(x.n, x.h, x.w, C), v = new_act(x.n, x.h, x.w, C);
    conv(w.q, g, q, 1, 0, nullptr, 0, nullptr);
    conv(w.k, g, k, 1, 0, nullptr, 0, nullptr);
    conv(w.v, g, v, 1, 0, nullptr, 0, nullptr);
    release(g);
    Act a = new_act(x.n, x.h, x.w, C);
    const long long bs = (long long)hw * C;
    attention(q.p, C, bs, k.p, C, bs, v.p, C, bs, a.p, C, bs, x.n, hw, hw, 1, C, nullptr, nullptr, nullptr, 0);
    release(q); release(k); release(v);
    conv(w.proj_out, a, y, 1, 0, nullptr, 0, &x);
    release(a);
}

// =============================================================================
// UNet driver
// =============================================================================
void Engine::unet_release() {
    for (void* p : us_.owned) pool_.free(p);
    us_ = UNetState{};
}

// Hoists everything that does not depend on the latent out of the step loop:
// the time-embedding MLP for every timestep (unet/mod.rs:115-118), each ResBlock's
// Linear(SiLU(emb)) (unet/mod.rs:718-719) and each cross-attention's K/V projection
// of the text context (unet/mod.rs:646-647), which the reference recomputes in all
// 2*n_steps forwards.
void Engine::unet_prepare(const float* ctx_packed, int nb, int t_max, const int* kv_len_host,
                          const std::vector<int>& ts) {
    unet_release();
    const int S = (int)ts.size(), mc = cfg_.model_channels, ed = 4 * mc, cd = cfg_.ctx_dim;
    us_.nb = nb; us_.t_max = t_max; us_.steps = S;
    us_.kv_len_host.assign(kv_len_host, kv_len_host + nb);
    auto own = [&](size_t bytes) { void* p = pool_.alloc(bytes); us_.owned.push_back(p); return p; };
    us_.kv_len_dev = (int*)own(nb * sizeof(int));
    int* t_dev = (int*)own(S * sizeof(int));
    SDMI_HIP(hipMemcpyAsync(us_.kv_len_dev, us_.kv_len_host.data(), nb * sizeof(int), hipMemcpyHostToDevice, stream_));
    SDMI_HIP(hipMemcpyAsync(t_dev, ts.data(), S * sizeof(int), hipMemcpyHostToDevice, stream_));
    SDMI_HIP(hipStreamSynchronize(stream_));  // host vectors may go away; once per call, outside the step loop

    Buf te(this, (size_t)S * mc * 4), e1(this, (size_t)S * ed * 4), e2(this, (size_t)S * ed * 4);
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_timestep_embedding(t_dev, S, mc, te.f(), stream_)); }
    count_kernel();
    gemm(te.f(), S, lin1_time_.bt, lin1_time_.bias, mc, ed, e1.f(), ed, nullptr, 0, /*dt=*/0);
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_silu(e1.f(), e1.f(), (long long)S * ed, stream_)); }
    count_kernel();
    gemm(e1.f(), S, lin2_time_.bt, lin2_time_.bias, ed, ed, e2.f(), ed, nullptr, 0, /*dt=*/0);
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_silu(e2.f(), e2.f(), (long long)S * ed, stream_)); }  // SiLU(emb), shared by all ResBlocks
    count_kernel();
    us_.temb.resize(res_list_.size());
    for (size_t i = 0; i < res_list_.size(); ++i) {
        const ResW& r = *res_list_[i];
        us_.temb[i] = (float*)own((size_t)S * r.cout * 4);
        gemm(e2.f(), S, r.lin_embed.bt, r.lin_embed.bias, ed, r.cout, us_.temb[i], r.cout, nullptr, 0, /*dt=*/0);
    }
    us_.kc.resize(st_list_.size());
    us_.vc.resize(st_list_.size());
    const float* ctx_e = ctx_packed;  // text context in the engine's storage type
    Buf ctx_h(this, bf16_ ? (size_t)nb * t_max * cd * 2 : 256);
    if (bf16_) {
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_f32_to_bf16(ctx_packed, ctx_h.p, (long long)nb * t_max * cd, stream_)); }
        count_kernel();
        ctx_e = ctx_h.f();
    }
    for (size_t i = 0; i < st_list_.size(); ++i) {
        const SpatialW& s = *st_list_[i];
        us_.kc[i] = (float*)own((size_t)nb * t_max * s.c * esz());
        us_.vc[i] = (float*)own((size_t)nb * t_max * s.c * esz());
        gemm(ctx_e, nb * t_max, s.attn2.k.bt, nullptr, cd, s.c, us_.kc[i], s.c, nullptr, 0);
        gemm(ctx_e, nb * t_max, s.attn2.v.bt, nullptr, cd, s.c, us_.vc[i], s.c, nullptr, 0);
    }
}

// UNet::forward (unet/mod.rs:109-143) on NHWC activations.
// Tensor::cat(vec![x, saved_inputs.pop()], 1) (unet/mod.rs:134) is not a copy here: the input buffer `cats[i]` of
// output block i ([x channels | skip channels]) is allocated when its skip is produced on the way down; input block
// 11 - i writes its result straight into the skip slice (the GEMM epilogue's row stride), output block i - 1 (or
// the middle block) writes the x slice, and output block i reads the whole buffer.
void Engine::unet_run(const float* x_nhwc, int nb, int step, float* out_nhwc, bool cfg_pair) {
    Range rng(this, "UNet::forward step " + std::to_string(step));
    const int H = cfg_.latent_h, W = cfg_.latent_w;
    Act x; x.p = const_cast<float*>(x_nhwc); x.n = nb; x.h = H; x.w = W; x.c = 4; x.dt = 0;  // latents stay fp32

    // every block's last kernel writes `y` (dense or a channel slice)
    auto run_block = [&](const UBlock& b, const Act& in, Act& y) {
        switch (b.kind) {
            case BK_CONV: conv(b.conv, in, y, 1, 0, nullptr, 0, nullptr); return;
            case BK_DOWN: conv_raw(b.conv, in, y, 2, 0); return;
            case BK_RES: res_block(b.res, in, y, step); return;
            case BK_RES_ST: {
                Act r = new_act(in.n, in.h, in.w, b.cout); res_block(b.res, in, r, step);
                spatial_transformer(b.st, r, y); release(r); return;
            }
            case BK_RES_UP: {   // (fp32 engine: the tensor between the block and its up-convolution exists only as planes)
                Act r = plane_gemm(b.cout, b.cout) ? new_act3(in.n, in.h, in.w, b.cout, 2) : new_act(in.n, in.h, in.w, b.cout);
                res_block(b.res, in, r, step);
                conv_raw(b.up, r, y, 1, 1); release(r); return;
            }
            case BK_RES_ST_UP: {
                Act r = new_act(in.n, in.h, in.w, b.cout); res_block(b.res, in, r, step);
                Act s = plane_gemm(b.cout, b.cout) ? new_act3(in.n, in.h, in.w, b.cout, 2) : new_act(in.n, in.h, in.w, b.cout);
                spatial_transformer(b.st, r, s); release(r);
                conv_raw(b.up, s, y, 1, 1); release(s); return;
            }
        }
        throw Error(SDMI_ERR_STATE, "bad block kind");
    };

    const int nblk = (int)in_blocks_.size();
    std::vector<Act> cats(nblk);
    for (int j = 0; j < nblk; ++j) {
        const UBlock& b = in_blocks_[j];
        Range rb(this, "input_block " + std::to_string(j));
        const int i = nblk - 1 - j;   // the output block that pops this skip (saved_inputs is a stack, unet/mod.rs:126,134)
        const int ctot = out_blocks_[i].cin, cskip = b.cout, cx = ctot - cskip;
        if (cx <= 0) throw Error(SDMI_ERR_STATE, "unet: block table inconsistent");
        const int ho = b.kind == BK_DOWN ? x.h / 2 : x.h, wo = b.kind == BK_DOWN ? x.w / 2 : x.w;
        // fp32 engine: the buffer also exists as planes -- the skip convolution of output block i (cin != cout) and the down-convolutions read those,
        // written by the epilogues (or split-K reduce kernels) of the GEMMs that produce the two halves
        const bool catp = plane_gemm(ctot, out_blocks_[i].cout) && cx % 32 == 0 && cskip % 32 == 0;
        cats[i] = catp ? new_act3(nb, ho, wo, ctot, 3) : new_act(nb, ho, wo, ctot);
        Act y = slice(cats[i], cx, cskip);
        if (cfg_pair && j == 1 && b.kind == BK_RES_ST && nb % 2 == 0) {
            // The two halves of a CFG step's batch -- uncond rows, then cond rows (stablediffusion/mod.rs:173-179: two forwards of the SAME x and t) -- are identical
            // until the text context first enters: conv_in, this block's ResBlock and its transformer up to the cross attention.  That prefix is computed ONCE on
            // the first half and duplicated in front of the cross attention (spatial_transformer): the same values the two forwards would produce, per sample
            // (GroupNorm and attention are per sample), for half the 64x64-level ResBlock convolutions and one of its five self-attentions less.  Option cfg_share=0
            // computes both halves.
            Act xh = x; xh.n = nb / 2;
            Act r = new_act(nb / 2, x.h, x.w, b.cout);
            res_block(b.res, xh, r, step);
            spatial_transformer(b.st, r, y);
            release(r);
        } else {
            run_block(b, x, y);
        }
        x = y;
    }
    {   // middle block: reads the last skip, writes the x slice of output block 0's input
        Act a = new_act(x.n, x.h, x.w, mid_res1_.cout); res_block(mid_res1_, x, a, step);
        Act b = new_act(x.n, x.h, x.w, mid_st_.c); spatial_transformer(mid_st_, a, b); release(a);
        Act c = slice(cats[0], 0, cats[0].c - x.c);
        if (c.c != mid_res2_.cout) throw Error(SDMI_ERR_STATE, "unet: middle block width mismatch");
        res_block(mid_res2_, b, c, step); release(b);
    }
    Act last{};
    for (int i = 0; i < nblk; ++i) {
        const UBlock& b = out_blocks_[i];
        Range rb(this, "output_block " + std::to_string(i));
        const bool up = b.kind == BK_RES_UP || b.kind == BK_RES_ST_UP;
        Act y;
        if (i + 1 < nblk) {
            y = slice(cats[i + 1], 0, cats[i + 1].c - in_blocks_[nblk - 2 - i].cout);
            if (y.c != b.cout || y.h != (cats[i].h << (up ? 1 : 0))) throw Error(SDMI_ERR_STATE, "unet: output block shape mismatch");
        } else {
            y = new_act(nb, cats[i].h << (up ? 1 : 0), cats[i].w << (up ? 1 : 0), b.cout);
            last = y;
        }
        run_block(b, cats[i], y);
        release(cats[i]);
    }
    Act gn = new_act(last.n, last.h, last.w, last.c);
    group_norm(unet_norm_out_, last, gn, true);
    release(last);
    Act out; out.p = out_nhwc; out.n = nb; out.h = H; out.w = W; out.c = 4; out.dt = 0;  // eps stays fp32
    conv(unet_conv_out_, gn, out, 1, 0, nullptr, 0, nullptr);
    release(gn);
}

// CLIP::forward (clip/mod.rs:56-75) with ResidualDecoderAttentionBlock (:110-114), MultiHeadSelfAttention
// (:158-180), MLP + QuickGELU (:207-226) and attn_decoder_mask (backend.rs:130-139).  fp32 in both precisions.
void Engine::clip_forward_dev(const int32_t* tokens, int n, int T, float* out) {
    if (!finalized_) throw Error(SDMI_ERR_STATE, "weights not finalized");
    if (!clip_ready_) throw Error(SDMI_ERR_STATE, "CLIP weights are not loaded (clip/... tensors; clip_layers > 0 in the config)");
    if (n <= 0 || T <= 0) throw Error(SDMI_ERR_INVALID, "clip_forward: n and seq_len must be positive");
    if (T > cfg_.clip_ctx) throw Error(SDMI_ERR_INVALID, "clip_forward: sequence longer than n_ctx");  // reference: slice panics
    const int C = cfg_.ctx_dim, H = cfg_.clip_heads;
    const long long M = (long long)n * T;
    Buf x(this, (size_t)M * C * 4), h(this, (size_t)M * C * 4), qkv(this, (size_t)M * 3 * C * 4), a(this, (size_t)M * C * 4);
    Buf f(this, (size_t)M * 4 * C * 4), mask(this, (size_t)T * T * 4);
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_clip_embed(tokens, clip_tok_, clip_pos_, x.f(), n, T, C, stream_)); }
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_causal_mask(mask.f(), T, stream_)); }
    count_kernel(); count_kernel();
    for (const ClipBlockW& b : clip_blocks_) {
        layer_norm(b.attn_ln, x.f(), M, h.f(), 0);
        gemm(h.f(), (int)M, b.q.bt, b.q.bias, C, 3 * C, qkv.f(), 3 * C, nullptr, 0, 0);       // query | key | value, one GEMM
        attention(qkv.f(), 3 * C, (long long)T * 3 * C, qkv.f() + C, 3 * C, (long long)T * 3 * C, qkv.f() + 2 * C, 3 * C,
                  (long long)T * 3 * C, a.f(), C, (long long)T * C, n, T, T, H, C / H, nullptr, nullptr, mask.f(), T, 0);
        gemm(a.f(), (int)M, b.out.bt, b.out.bias, C, C, x.f(), C, x.f(), C, 0);               // x += out(attn)
        layer_norm(b.mlp_ln, x.f(), M, h.f(), 0);
        gemm(h.f(), (int)M, b.fc1.bt, b.fc1.bias, C, 4 * C, f.f(), 4 * C, nullptr, 0, 0);
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_quick_gelu(f.f(), M * 4 * C, stream_)); }
        count_kernel();
        gemm(f.f(), (int)M, b.fc2.bt, b.fc2.bias, 4 * C, C, x.f(), C, x.f(), C, 0);           // x += fc2(gelu(fc1))
    }
    layer_norm(clip_ln_, x.f(), M, out, 0);
}

void Engine::unet_forward_dev(const float* x_nchw, int t, const float* context, int n, int T, float* out_nchw) {
    if (!finalized_) throw Error(SDMI_ERR_STATE, "weights not finalized");
    if (n <= 0 || T <= 0) throw Error(SDMI_ERR_INVALID, "unet_forward: n and T must be positive");
    check_batch(n);
    const int H = cfg_.latent_h, W = cfg_.latent_w;
    std::vector<int> kv(n, T), ts(1, t);
    unet_prepare(context, n, T, kv.data(), ts);
    Buf xin(this, (size_t)n * H * W * 4 * 4), xout(this, (size_t)n * H * W * 4 * 4);
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nchw_to_nhwc(x_nchw, xin.f(), n, 4, H, W, 1.0f, stream_)); }
    count_kernel();
    unet_run(xin.f(), n, 0, xout.f());
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nhwc_to_nchw(xout.f(), out_nchw, n, 4, H, W, stream_)); }
    count_kernel();
    unet_release();
}

// StableDiffusion::sample_latent + forward_diffuser (stablediffusion/mod.rs:102-192)
void Engine::sample_latent_dev(const float* context, int n, int T, const float* uncond, int Tu, double scale,
                               size_t n_steps, const float* init_latent, float* latent_out) {
    if (!finalized_) throw Error(SDMI_ERR_STATE, "weights not finalized");
    if (n <= 0 || T <= 0 || Tu <= 0) throw Error(SDMI_ERR_INVALID, "sample_latent: n, T, Tu must be positive");
    check_batch(n);
    const size_t total = alphas_.size();
    if (n_steps == 0 || n_steps > total) throw Error(SDMI_ERR_INVALID, "sample_latent: n_steps out of range");
    const int H = cfg_.latent_h, W = cfg_.latent_w, cd = cfg_.ctx_dim;
    const int nb = 2 * n, t_max = std::max(T, Tu);
    const size_t step_size = total / n_steps;                       // :111
    std::vector<int> ts;
    for (long long t = (long long)total - 1; t >= 0; t -= (long long)step_size) ts.push_back((int)t);  // :123

    // packed context [2n][t_max][cd]: rows 0..n-1 = uncond (broadcast, :173-177), n..2n-1 = cond
    Buf ctx(this, (size_t)nb * t_max * cd * 4);
    SDMI_HIP(hipMemsetAsync(ctx.p, 0, (size_t)nb * t_max * cd * 4, stream_));
    for (int b = 0; b < n; ++b)
        SDMI_HIP(hipMemcpyAsync(ctx.f() + (size_t)b * t_max * cd, uncond, (size_t)Tu * cd * 4, hipMemcpyDeviceToDevice, stream_));
    for (int b = 0; b < n; ++b)
        SDMI_HIP(hipMemcpyAsync(ctx.f() + (size_t)(n + b) * t_max * cd, context + (size_t)b * T * cd, (size_t)T * cd * 4,
                                hipMemcpyDeviceToDevice, stream_));
    std::vector<int> kv(nb);
    for (int b = 0; b < n; ++b) { kv[b] = Tu; kv[n + b] = T; }
    unet_prepare(ctx.f(), nb, t_max, kv.data(), ts);

    const long long per_half = (long long)n * H * W * 4;
    Buf latent(this, per_half * 4), unet_in(this, 2 * per_half * 4), eps(this, 2 * per_half * 4);
    SDMI_HIP(launch_nchw_to_nhwc(init_latent, latent.f(), n, 4, H, W, 1.0f, stream_));
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_dup_latent(latent.f(), unet_in.f(), per_half, stream_)); }
    count_kernel(); count_kernel();
    for (size_t s = 0; s < ts.size(); ++s) {
        const size_t t = (size_t)ts[s];
        const double cur = (double)alphas_[t];                                           // :124-129
        const double prev = t >= step_size ? (double)alphas_[t - step_size] : 1.0;       // :131-140
        DdimCoef c{};
        c.scale = (float)scale;
        c.sqrt_noise = (float)std::sqrt(1.0 - cur);                                      // :142
        c.sqrt_cur = (float)std::sqrt(cur);
        c.sqrt_prev = (float)std::sqrt(prev);
        c.dir_coef = (float)std::sqrt(1.0 - prev - 0.0);                                 // :153 (sigma = 0)
        unet_run(unet_in.f(), nb, (int)s, eps.f(), opt_cfg_share_ != 0);   // unet_in = [latent | latent]: a CFG pair
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_cfg_ddim(eps.f(), latent.f(), unet_in.f(), per_half, c, stream_)); }
        count_kernel();
    }
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nhwc_to_nchw(latent.f(), latent_out, n, 4, H, W, stream_)); }
    count_kernel();
    unet_release();
}

// Autoencoder::decode_latent (autoencoder/mod.rs:68-71) -> Decoder::forward (:205-217)
void Engine::decode_one(const float* z_nhwc, int n, Act& img) {
    Range rng(this, "Decoder::forward");
    const int H = cfg_.latent_h, W = cfg_.latent_w, vc = cfg_.vae_ch;
    Act z; z.p = const_cast<float*>(z_nhwc); z.n = n; z.h = H; z.w = W; z.c = 4; z.dt = 0;
    Act pq = new_act(n, H, W, 4, /*dt=*/0);  // the two Cin = 4 layers run on the fp32 kernel in both precisions
    conv(post_quant_, z, pq, 1, 0, nullptr, 0, nullptr);
    Act x = new_act(n, H, W, 4 * vc);
    conv(dec_conv_in_, pq, x, 1, 0, nullptr, 0, nullptr);
    release(pq);
    {   // Mid (:457-462)
        Act a = new_act(n, H, W, 4 * vc); res_block(dec_mid1_, x, a, 0); release(x);
        Act b = new_act(n, H, W, 4 * vc); vae_attn(dec_attn_, a, b); release(a);
        Act c = new_act(n, H, W, 4 * vc); res_block(dec_mid2_, b, c, 0); release(b);
        x = c;
    }
    for (int i = 0; i < 4; ++i) {  // DecoderBlock::forward (:308-323)
        const DecBlockW& b = dec_blocks_[i];
        for (int r = 0; r < 3; ++r) {
            // fp32 engine: the block's last tensor is read by the up-convolution only -> planes only
            Act y = (r == 2 && b.has_up && plane_gemm(b.cout, b.cout)) ? new_act3(x.n, x.h, x.w, b.cout, 2) : new_act(x.n, x.h, x.w, b.cout);
            res_block(b.res[r], x, y, 0);
            release(x);
            x = y;
        }
        if (b.has_up) {
            // the next block's first ResnetBlock has a 1x1 shortcut (cin != cout): it reads the up-convolution's output as planes too
            const bool both = i + 1 < 4 && dec_blocks_[i + 1].res[0].has_skip && plane_gemm(b.cout, dec_blocks_[i + 1].cout);
            Act y = both ? new_act3(x.n, x.h * 2, x.w * 2, b.cout, 3) : new_act(x.n, x.h * 2, x.w * 2, b.cout);
            conv_raw(b.upsampler, x, y, 1, 1);
            release(x);
            x = y;
        }
    }
    Act gn = new_act(x.n, x.h, x.w, x.c);
    group_norm(dec_norm_out_, x, gn, true);
    release(x);
    conv(dec_conv_out_, gn, img, 1, 0, nullptr, 0, nullptr);
    release(gn);
}

// Autoencoder::encode_image (autoencoder/mod.rs:60-66): Encoder::forward (:133-144) -> quant_conv -> channels 0..3
// (the posterior mean; the reference does not sample).  One image at a time, like the decoder.
void Engine::encode_image_dev(const float* img_nchw, int n, float* latent_nchw) {
    if (!finalized_) throw Error(SDMI_ERR_STATE, "weights not finalized");
    if (!enc_ready_) throw Error(SDMI_ERR_STATE, "VAE encoder weights are not loaded (autoencoder/encoder/..., autoencoder/quant_conv)");
    if (n <= 0) throw Error(SDMI_ERR_INVALID, "encode_image: n must be positive");
    check_batch(n);
    const int H = 8 * cfg_.latent_h, W = 8 * cfg_.latent_w;
    const size_t img_elems = (size_t)3 * H * W, lat_elems = (size_t)4 * cfg_.latent_h * cfg_.latent_w;
    for (int i = 0; i < n; ++i) {
        Act rgb = new_act(1, H, W, 4, /*dt=*/0);
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nchw3_to_nhwc4(img_nchw + i * img_elems, rgb.p, 1, H, W, stream_)); }
        count_kernel();
        Act x = new_act(1, H, W, enc_conv_in_.cout);
        conv(enc_conv_in_, rgb, x, 1, 0, nullptr, 0, nullptr);
        release(rgb);
        for (int bi = 0; bi < 4; ++bi) {  // EncoderBlock::forward (:257-265)
            const EncBlockW& b = enc_blocks_[bi];
            for (int r = 0; r < 2; ++r) {
                Act y = new_act(x.n, x.h, x.w, b.cout);
                res_block(b.res[r], x, y, 0);
                release(x);
                x = y;
            }
            if (b.has_down) {
                Act y = new_act(x.n, x.h / 2, x.w / 2, b.cout);
                conv(b.down, x, y, 2, 0, nullptr, 0, nullptr, /*pad_br=*/true);
                release(x);
                x = y;
            }
        }
        {   // Mid (:457-462)
            Act a = new_act(x.n, x.h, x.w, x.c); res_block(enc_mid1_, x, a, 0); release(x);
            Act b = new_act(a.n, a.h, a.w, a.c); vae_attn(enc_attn_, a, b); release(a);
            Act c = new_act(b.n, b.h, b.w, b.c); res_block(enc_mid2_, b, c, 0); release(b);
            x = c;
        }
        Act gn = new_act(x.n, x.h, x.w, x.c);
        group_norm(enc_norm_out_, x, gn, true);
        release(x);
        Act m8 = new_act(gn.n, gn.h, gn.w, 8, /*dt=*/0);   // moments stay fp32 in both precisions
        conv(enc_conv_out_, gn, m8, 1, 0, nullptr, 0, nullptr);
        release(gn);
        Act q8 = new_act(m8.n, m8.h, m8.w, 8, /*dt=*/0);
        conv(quant_conv_, m8, q8, 1, 0, nullptr, 0, nullptr);
        release(m8);
        // latent.slice([0..n, 0..4]): the first 4 of the 8 channels, NHWC8 -> NCHW4
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nhwc_to_nchw_slice(q8.p, latent_nchw + i * lat_elems, 1, 8, 4, q8.h, q8.w, stream_)); }
        count_kernel();
        release(q8);
    }
}

// in_scale = 1/0.18215 for latent_to_image (stablediffusion/mod.rs:71), 1 for decode_latent.
// Exactly one of img_nchw / rgb_u8 is written.
void Engine::decode_latent_dev(const float* latent_nchw, int n, float in_scale, float* img_nchw, uint8_t* rgb_u8) {
    if (!finalized_) throw Error(SDMI_ERR_STATE, "weights not finalized");
    if (n <= 0) throw Error(SDMI_ERR_INVALID, "decode: n must be positive");
    check_batch(n);
    const int H = cfg_.latent_h, W = cfg_.latent_w;
    const size_t lat_elems = (size_t)4 * H * W, img_elems = (size_t)3 * 64 * H * W;
    for (int i = 0; i < n; ++i) {  // one image at a time: peak activations are ~1 GB per image at 512x512
        Buf z(this, lat_elems * 4);
        { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nchw_to_nhwc(latent_nchw + i * lat_elems, z.f(), 1, 4, H, W, in_scale, stream_)); }
        count_kernel();
        Act img = new_act(1, 8 * H, 8 * W, 3, /*dt=*/0);  // RGB stays fp32 (conv_out writes fp32 in both precisions)
        decode_one(z.f(), 1, img);
        if (rgb_u8) { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_image_to_u8(img.p, rgb_u8 + i * img_elems, (long long)img_elems, stream_)); }
        else { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_nhwc_to_nchw(img.p, img_nchw + i * img_elems, 1, 3, 8 * H, 8 * W, stream_)); }
        count_kernel();
        release(img);
    }
}

void Engine::qkv_attention_dev(const float* q, const float* k, const float* v, const float* mask, int mask_ld, int n,
                               int nq, int nk, int n_state, int n_head, float* out) {
    if (n <= 0 || nq <= 0 || nk <= 0 || n_head <= 0 || n_state % n_head) throw Error(SDMI_ERR_INVALID, "qkv_attention: bad shape");
    if (mask && mask_ld < nk) throw Error(SDMI_ERR_INVALID, "qkv_attention: mask_ld < nk");
    const long long qe = (long long)n * nq * n_state, ke = (long long)n * nk * n_state;
    if (bf16_ && !mask) {  // precision = 1: the boundary is fp32, the kernel sees bf16 tensors
        Buf qh(this, qe * 2), kh(this, ke * 2), vh(this, ke * 2), oh(this, qe * 2);
        const int dh = n_state / n_head;
        if (q_prescaled(1, dh)) SDMI_HIP(launch_f32_to_bf16_scaled(q, qh.p, qe, attn_bf16_q_scale(dh), stream_));   // Engine::attention's q convention
        else SDMI_HIP(launch_f32_to_bf16(q, qh.p, qe, stream_));
        SDMI_HIP(launch_f32_to_bf16(k, kh.p, ke, stream_));
        SDMI_HIP(launch_f32_to_bf16(v, vh.p, ke, stream_));
        attention(qh.f(), n_state, (long long)nq * n_state, kh.f(), n_state, (long long)nk * n_state, vh.f(), n_state,
                  (long long)nk * n_state, oh.f(), n_state, (long long)nq * n_state, n, nq, nk, n_head, n_state / n_head,
                  nullptr, nullptr, nullptr, 0, 1, nullptr, q_prescaled(1, dh));
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(oh.p, out, (int)((long long)n * nq), n_state, 1, 1, stream_));
        return;
    }
    if (plane_gemm(n_state, n_state) && attn_supported_head_dim(n_state / n_head)) {   // option gemm_planes: the kernels' plane-writing epilogue, joined back
        Buf o3(this, (size_t)n * nq * (n_state / 32) * 192);
        attention(q, n_state, (long long)nq * n_state, k, n_state, (long long)nk * n_state, v, n_state,
                  (long long)nk * n_state, nullptr, n_state, (long long)nq * n_state, n, nq, nk, n_head, n_state / n_head,
                  nullptr, nullptr, mask, mask_ld, 0, o3.p);
        SDMI_HIP(launch_join3_rows(o3.p, out, (long long)n * nq, n_state, (long long)(n_state / 32) * 192, n_state, stream_));
        return;
    }
    attention(q, n_state, (long long)nq * n_state, k, n_state, (long long)nk * n_state, v, n_state,
              (long long)nk * n_state, out, n_state, (long long)nq * n_state, n, nq, nk, n_head, n_state / n_head,
              nullptr, nullptr, mask, mask_ld, 0);
}

// =============================================================================
// operator-level entry points (device pointers, reference layouts)
// =============================================================================
void Engine::op_group_norm(const float* x, const float* gamma, const float* beta, int n, int c, int h, int w,
                           int groups, float eps, bool silu, float* out) {
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || groups <= 0 || c % groups || c % 4) throw Error(SDMI_ERR_INVALID, "group_norm: bad shape");
    const int dt = (bf16_ && c % 8 == 0) ? 1 : 0;
    Act a = new_act(n, h, w, c, dt), b = new_act(n, h, w, c, dt);
    if (dt) {
        SDMI_HIP(launch_nchw_f32_to_nhwc_bf16(x, a.p, n, c, h, w, 1.0f, stream_));
        Buf part(this, gn_partials_bytes_bf16(n, h * w, c, gn_tune_));
        SDMI_HIP(launch_group_norm_bf16(a.p, b.p, gamma, beta, n, h * w, c, c, groups, eps, silu, part.p, stream_, gn_tune_));
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(b.p, out, n, c, h, w, stream_));
    } else {
        SDMI_HIP(launch_nchw_to_nhwc(x, a.p, n, c, h, w, 1.0f, stream_));
        Buf part(this, gn_partials_bytes(n, h * w, c, opt_gn32_min_wgs_));
        if (plane_gemm(c, c)) {   // option gemm_planes: the plane-writing form of the kernel, joined back to fp32 (exact)
            Buf y3(this, (size_t)n * h * w * (c / 32) * 192);
            SDMI_HIP(launch_group_norm_planes(a.p, y3.p, gamma, beta, n, h * w, c, c, groups, eps, silu, part.p, stream_, opt_gn32_min_wgs_));
            SDMI_HIP(launch_join3_rows(y3.p, b.p, (long long)n * h * w, c, (long long)(c / 32) * 192, c, stream_));
        } else {
            SDMI_HIP(launch_group_norm(a.p, b.p, gamma, beta, n, h * w, c, c, groups, eps, silu, part.p, stream_, opt_gn32_min_wgs_));
        }
        SDMI_HIP(launch_nhwc_to_nchw(b.p, out, n, c, h, w, stream_));
    }
    release(a); release(b);
}

// GroupNorm(+SiLU) with MXFP8 output (precision = 2), returned dequantised: out [n,c,h,w] fp32
void Engine::op_group_norm_fp8(const float* x, const float* gamma, const float* beta, int n, int c, int h, int w, int groups, float eps,
                               bool silu, float* out) {
    if (!fp8_) throw Error(SDMI_ERR_STATE, "group_norm_fp8 needs a precision = 2 context");
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || groups <= 0 || c % groups || c % 32) throw Error(SDMI_ERR_INVALID, "group_norm_fp8: bad shape");
    Act a = new_act(n, h, w, c, 1);
    SDMI_HIP(launch_nchw_f32_to_nhwc_bf16(x, a.p, n, c, h, w, 1.0f, stream_));
    ActQ q = new_actq(n, h, w, c);
    Buf part(this, gn_partials_bytes_bf16(n, h * w, c, gn_tune_));
    SDMI_HIP(launch_group_norm_fp8(a.p, q.q, q.s, gamma, beta, n, h * w, c, c, groups, eps, silu, part.p, stream_, &gn_tune_));
    Act d = new_act(n, h, w, c, 0);
    SDMI_HIP(launch_dequant_fp8(q.q, q.s, d.p, d.rows(), c, stream_));
    SDMI_HIP(launch_nhwc_to_nchw(d.p, out, n, c, h, w, stream_));
    release(a); release(q); release(d);
}

void Engine::op_layer_norm(const float* x, const float* gamma, const float* beta, int rows, int c, float eps, float* out) {
    if (rows <= 0 || c <= 0) throw Error(SDMI_ERR_INVALID, "layer_norm: bad shape");
    if (fp8_ && opt_fp8_ops_ && c % 32 == 0) {   // option fp8_ops (tests): the quantising LayerNorm of the fp8_linear path, dequantised
        Buf xh(this, (size_t)rows * c * 2);
        SDMI_HIP(launch_f32_to_bf16(x, xh.p, (long long)rows * c, stream_));
        ActQ q = new_rowsq(rows, c);
        SDMI_HIP(launch_layer_norm_fp8(xh.p, q.q, q.s, gamma, beta, rows, c, eps, stream_));
        SDMI_HIP(launch_dequant_fp8(q.q, q.s, out, rows, c, stream_));
        release(q);
        return;
    }
    if (bf16_ && c % 8 == 0) {
        Buf xh(this, (size_t)rows * c * 2), yh(this, (size_t)rows * c * 2);
        SDMI_HIP(launch_f32_to_bf16(x, xh.p, (long long)rows * c, stream_));
        SDMI_HIP(launch_layer_norm_bf16(xh.p, yh.p, gamma, beta, rows, c, eps, stream_));
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(yh.p, out, rows, c, 1, 1, stream_));
        return;
    }
    if (plane_gemm(c, c)) {
        Buf y3(this, (size_t)rows * (c / 32) * 192);
        SDMI_HIP(launch_layer_norm_planes(x, y3.p, gamma, beta, rows, c, eps, stream_));
        SDMI_HIP(launch_join3_rows(y3.p, out, rows, c, (long long)(c / 32) * 192, c, stream_));
        return;
    }
    SDMI_HIP(launch_layer_norm(x, out, gamma, beta, rows, c, eps, stream_));
}

void Engine::op_conv2d(const float* x, const float* wt, const float* bias, int n, int cin, int h, int wd, int cout,
                       int k, int stride, int pad, int ups, float* out) {
    if (!(k == 1 || k == 3) || pad != (k == 3 ? 1 : 0) || !(stride == 1 || stride == 2))
        throw Error(SDMI_ERR_UNSUPPORTED, "conv2d: only 3x3 pad 1 / 1x1 pad 0, stride 1|2 are on the hot path");
    if (!(cin % 32 == 0 || (cin < 32 && cin % 4 == 0))) throw Error(SDMI_ERR_UNSUPPORTED, "conv2d: Cin must be a multiple of 32, or < 32 and a multiple of 4");
    if (fp8_ && opt_fp8_convs_ && k == 3 && stride == 1 && !ups && cin % 32 == 0 && cout % 8 == 0) {
        // precision = 2 mirrors the model's ResBlock convs: MXFP8 input (quantised here from the fp32 argument; the model
        // gets it from the fused GroupNorm), MXFP8 weight, bf16 output
        const int cp = (cin + 127) / 128 * 128;
        ConvW w; w.cin = cin; w.cout = cout; w.k = 3; w.dt = 1; w.bias = const_cast<float*>(bias);
        Buf bt8(this, (size_t)cout * cp * 9), bs8(this, (size_t)cout * cp * 9 / 32);
        SDMI_HIP(launch_pack_conv_weight_fp8(wt, bt8.p, bs8.p, cout, cin, 3, 3, stream_));
        w.bt8 = bt8.f(); w.bs8 = bs8.f();
        Act a32 = new_act(n, h, wd, cin, 0);
        SDMI_HIP(launch_nchw_to_nhwc(x, a32.p, n, cin, h, wd, 1.0f, stream_));
        ActQ q = new_actq(n, h, wd, cin);
        SDMI_HIP(launch_quantize_fp8(a32.p, q.q, q.s, a32.rows(), cin, stream_));
        release(a32);
        Act y = new_act(n, h, wd, cout, 1);
        conv_fp8(w, q, y, nullptr, nullptr);
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(y.p, out, n, cout, h, wd, stream_));
        release(q); release(y);
        return;
    }
    ConvW w; w.cin = cin; w.cout = cout; w.k = k;
    // precision = 1 mirrors the model: Cin % 64 == 0 -> bf16 kernel, Cin < 32 -> fp32 kernel emitting bf16,
    // <= 4 output channels (eps / RGB heads) -> fp32 output
    w.dt = (bf16_ && cin % 64 == 0) ? 1 : 0;
    if (bf16_ && !w.dt && cin >= 32) throw Error(SDMI_ERR_UNSUPPORTED, "bf16 conv2d: Cin must be a multiple of 64 (or < 32)");
    Buf bt(this, (size_t)cout * cin * k * k * 4);
    if (w.dt) SDMI_HIP(launch_pack_conv_weight_bf16(wt, bt.p, cout, cin, k, k, stream_));
    else SDMI_HIP(launch_pack_conv_weight(wt, bt.f(), cout, cin, k, k, stream_));
    TempSplit planes(this, bt.f(), w.dt ? 0 : cout, (long long)cin * k * k);
    w.bt = bt.f(); w.bias = const_cast<float*>(bias);
    Act a = new_act(n, h, wd, cin, w.dt);
    if (w.dt) SDMI_HIP(launch_nchw_f32_to_nhwc_bf16(x, a.p, n, cin, h, wd, 1.0f, stream_));
    else SDMI_HIP(launch_nchw_to_nhwc(x, a.p, n, cin, h, wd, 1.0f, stream_));
    const int hin = h << ups, win = wd << ups;
    const int ho = (hin + 2 * pad - k) / stride + 1, wo = (win + 2 * pad - k) / stride + 1;
    Act y = new_act(n, ho, wo, cout, (bf16_ && cout > 4) ? 1 : 0);
    // option op_resid (tests): out = conv(x) + x through the GEMM's residual epilogue, where the shapes allow it
    const bool with_resid = opt_op_resid_ && cin == cout && stride == 1 && !ups && a.dt == y.dt;
    conv(w, a, y, stride, ups, nullptr, 0, with_resid ? &a : nullptr);
    if (y.dt) SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(y.p, out, n, cout, ho, wo, stream_));
    else SDMI_HIP(launch_nhwc_to_nchw(y.p, out, n, cout, ho, wo, stream_));
    release(a); release(y);
}

void Engine::op_linear(const float* x, const float* wt, const float* bias, int rows, int cin, int cout, float* out) {
    if (fp8_ && opt_fp8_ops_ && cin % 32 == 0 && cout % 8 == 0) {   // option fp8_ops (tests): the Linear layer as the fp8_linear path runs it
        const size_t kp = (size_t)(cin + 127) / 128 * 128;
        Buf xh(this, (size_t)rows * cin * 2), yh(this, (size_t)rows * cout * 2), w8(this, (size_t)cout * kp), s8(this, (size_t)cout * kp / 32);
        SDMI_HIP(launch_f32_to_bf16(x, xh.p, (long long)rows * cin, stream_));
        SDMI_HIP(launch_pack_linear_weight_fp8(wt, w8.p, s8.p, cin, cout, stream_));
        Act xa; xa.p = xh.f(); xa.n = 1; xa.h = 1; xa.w = rows; xa.c = cin; xa.dt = 1;
        ActQ q = new_rowsq(rows, cin);
        quantize(xa, q);
        LinW lw; lw.cin = cin; lw.cout = cout; lw.dt = 1; lw.bias = const_cast<float*>(bias); lw.bt8 = w8.f(); lw.bs8 = s8.f();
        gemm_fp8(q, lw, cout, yh.p, cout, nullptr, 0);
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(yh.p, out, rows, cout, 1, 1, stream_));
        release(q);
        return;
    }
    Buf bt(this, (size_t)cin * cout * 4);
    if (bf16_ && cin % 64 == 0) {
        Buf xh(this, (size_t)rows * cin * 2), yh(this, (size_t)rows * cout * 2);
        SDMI_HIP(launch_pack_linear_weight_bf16(wt, bt.p, cin, cout, stream_));
        SDMI_HIP(launch_f32_to_bf16(x, xh.p, (long long)rows * cin, stream_));
        const bool with_resid = opt_op_resid_ && cin == cout;   // option op_resid (tests): out = x W + b + x through the residual epilogue
        gemm(xh.f(), rows, bt.f(), bias, cin, cout, yh.f(), cout, with_resid ? xh.f() : nullptr, with_resid ? cin : 0, 1);
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(yh.p, out, rows, cout, 1, 1, stream_));
        return;
    }
    SDMI_HIP(launch_pack_linear_weight(wt, bt.f(), cin, cout, stream_));
    TempSplit planes(this, bt.f(), cout, cin);
    const bool with_resid = opt_op_resid_ && cin == cout;
    gemm(x, rows, bt.f(), bias, cin, cout, out, cout, with_resid ? x : nullptr, with_resid ? cin : 0, 0);
}

void Engine::op_geglu_forward(const float* x, const float* wt, const float* bias, int rows, int cin, int hidden, float* out) {
    Buf bt(this, (size_t)cin * 2 * hidden * 4);
    if (bf16_ && cin % 64 == 0 && hidden % 8 == 0) {
        Buf xh(this, (size_t)rows * cin * 2), yh(this, (size_t)rows * hidden * 2);
        SDMI_HIP(launch_pack_linear_weight_bf16(wt, bt.p, cin, 2 * hidden, stream_));
        SDMI_HIP(launch_f32_to_bf16(x, xh.p, (long long)rows * cin, stream_));
        gemm_geglu(xh.f(), rows, bt.f(), bias, cin, hidden, yh.f(), 1);
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(yh.p, out, rows, hidden, 1, 1, stream_));
        return;
    }
    SDMI_HIP(launch_pack_linear_weight(wt, bt.f(), cin, 2 * hidden, stream_));
    TempSplit planes(this, bt.f(), 2 * hidden, cin);
    if (opt_geglu_fuse_ >= 7 && plane_gemm(cin, hidden) && hidden % 32 == 0) {
        // tests: the model's plane form -- x as planes in, the gated result as planes out (joined back exactly) -- through the plane GEMM's fused gate;
        // geglu_fuse = 7: the tile the table / cost model picks, 8: also the fp32 result from the same launch (both outputs of the epilogue)
        Buf x3(this, (size_t)rows * (cin / 32) * 192), o3(this, (size_t)rows * (hidden / 32) * 192);
        SDMI_HIP(launch_split3_rows(x, x3.p, rows, cin, cin, (long long)(cin / 32) * 192, stream_));
        if (opt_geglu_fuse_ == 8) {
            Buf o32(this, (size_t)rows * hidden * 4);
            gemm_geglu(nullptr, rows, bt.f(), bias, cin, hidden, o32.f(), 0, x3.p, o3.p);
            SDMI_HIP(hipMemcpyAsync(out, o32.p, (size_t)rows * hidden * 4, hipMemcpyDeviceToDevice, stream_));   // the fp32 output of a launch that writes both
        } else {
            gemm_geglu(nullptr, rows, bt.f(), bias, cin, hidden, nullptr, 0, x3.p, o3.p);
            SDMI_HIP(launch_join3_rows(o3.p, out, rows, hidden, (long long)(hidden / 32) * 192, hidden, stream_));
        }
        return;
    }
    gemm_geglu(x, rows, bt.f(), bias, cin, hidden, out, 0);
}

void Engine::op_geglu(const float* proj, int rows, int hidden, float* out) {
    if (fp8_ && opt_fp8_ops_ && hidden % 32 == 0) {   // option fp8_ops (tests): the quantising gate of the fp8_linear path, dequantised
        Buf ph(this, (size_t)rows * 2 * hidden * 2);
        SDMI_HIP(launch_f32_to_bf16(proj, ph.p, (long long)rows * 2 * hidden, stream_));
        ActQ q = new_rowsq(rows, hidden);
        SDMI_HIP(launch_geglu_fp8(ph.p, q.q, q.s, rows, hidden, stream_));
        SDMI_HIP(launch_dequant_fp8(q.q, q.s, out, rows, hidden, stream_));
        release(q);
        return;
    }
    if (bf16_ && hidden % 8 == 0) {
        Buf ph(this, (size_t)rows * 2 * hidden * 2), oh(this, (size_t)rows * hidden * 2);
        SDMI_HIP(launch_f32_to_bf16(proj, ph.p, (long long)rows * 2 * hidden, stream_));
        SDMI_HIP(launch_geglu_bf16(ph.p, oh.p, rows, hidden, stream_));
        SDMI_HIP(launch_nhwc_bf16_to_nchw_f32(oh.p, out, rows, hidden, 1, 1, stream_));
        return;
    }
    if (plane_gemm(hidden, hidden)) {
        Buf y3(this, (size_t)rows * (hidden / 32) * 192);
        SDMI_HIP(launch_geglu_planes(proj, y3.p, rows, hidden, stream_));
        SDMI_HIP(launch_join3_rows(y3.p, out, rows, hidden, (long long)(hidden / 32) * 192, hidden, stream_));
        return;
    }
    SDMI_HIP(launch_geglu(proj, out, rows, hidden, stream_));
}

void Engine::op_timestep_embedding(int t, int dim, float* out) {
    Buf td(this, sizeof(int));
    SDMI_HIP(hipMemcpyAsync(td.p, &t, sizeof(int), hipMemcpyHostToDevice, stream_));
    SDMI_HIP(hipStreamSynchronize(stream_));
    { ProfScope ps_o(this, PC_OTHER); SDMI_HIP(launch_timestep_embedding((const int*)td.p, 1, dim, out, stream_)); }
    SDMI_HIP(hipStreamSynchronize(stream_));
}

// option gemm_probe: what the diagnostic instantiation of a plane GEMM stored (24 words per workgroup: kernels.hpp ConvGemm::probe), summarised on stderr
void Engine::probe_report(void* pb_dev, size_t kMaxBlocks, int n, int cin, int h, int w, int cout, int k, int tile_cfg, int splitk) {
    std::vector<unsigned long long> hb(kMaxBlocks * 24);
    SDMI_HIP(hipMemcpyAsync(hb.data(), pb_dev, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    SDMI_HIP(hipStreamSynchronize(stream_));
    std::vector<double> pro, loop, epi, tot, wait_frac, mhz;
    unsigned long long first = ~0ull, last = 0, last_start = 0;
    for (size_t b = 0; b < kMaxBlocks; ++b) {
        const unsigned long long* d = &hb[24 * b];
        if (!d[0] || !d[3]) continue;
        pro.push_back((d[1] - d[0]) * 0.01); loop.push_back((d[2] - d[1]) * 0.01); epi.push_back((d[3] - d[2]) * 0.01); tot.push_back((d[3] - d[0]) * 0.01);
        first = std::min(first, d[0]); last = std::max(last, d[3]); last_start = std::max(last_start, d[0]);
        for (int wv = 0; wv < 8; ++wv)
            if (d[4 + 2 * wv]) {
                wait_frac.push_back((double)d[5 + 2 * wv] / (double)d[4 + 2 * wv]);
                if (d[2] > d[1]) mhz.push_back((double)d[4 + 2 * wv] / ((d[2] - d[1]) * 0.01));
            }
    }
    auto q = [](std::vector<double>& v, double f) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
    std::fprintf(stderr, "gemm_probe n=%d cin=%d %dx%d cout=%d k=%d tile=%d splitk=%d cold=%d: %zu workgroups; us min/median/max: first tile %.2f/%.2f/%.2f, "
                         "k loop %.2f/%.2f/%.2f, epilogue %.2f/%.2f/%.2f, workgroup %.2f/%.2f/%.2f; first entry -> last entry %.2f, first entry -> last exit %.2f; "
                         "per wave: share of the k loop waiting at the per-tile barrier %.3f/%.3f/%.3f, shader clock in the k loop %.0f/%.0f/%.0f MHz\n",
                 n, cin, h, w, cout, k, tile_cfg, splitk, opt_bench_cold_, tot.size(), q(pro, 0), q(pro, 0.5), q(pro, 1), q(loop, 0), q(loop, 0.5), q(loop, 1),
                 q(epi, 0), q(epi, 0.5), q(epi, 1), q(tot, 0), q(tot, 0.5), q(tot, 1), (last_start - first) * 0.01, (last - first) * 0.01,
                 q(wait_frac, 0), q(wait_frac, 0.5), q(wait_frac, 1), q(mhz, 0), q(mhz, 0.5), q(mhz, 1));
}

double Engine::bench_conv(int n, int cin, int h, int w, int cout, int k, int stride, int ups, int tile_cfg, int splitk,
                          int iters) {
    SDMI_HIP(hipSetDevice(cfg_.device));
    const int pad = k == 3 ? 1 : 0;
    const int hin = h << ups, win = w << ups;
    const int ho = (hin + 2 * pad - k) / stride + 1, wo = (win + 2 * pad - k) / stride + 1;
    if (fp8_ && opt_fp8_convs_ && k == 3 && stride == 1 && !ups && cin % 32 == 0 && cout % 8 == 0) {
        // precision = 2: time the MXFP8 kernel on a pre-quantised activation (what the fused GroupNorm hands it)
        const int cp = (cin + 127) / 128 * 128;
        Act x32 = new_act(n, h, w, cin, 0), y = new_act(n, h, w, cout, 1);
        Buf w32(this, (size_t)cout * cin * 9 * 4), bt8(this, (size_t)cout * cp * 9), bs8(this, (size_t)cout * cp * 9 / 32), bias(this, (size_t)cout * 4);
        SDMI_HIP(launch_fill_normal(x32.p, (long long)x32.rows() * cin, 11, stream_));
        SDMI_HIP(launch_fill_normal(w32.f(), (long long)cout * cin * 9, 12, stream_));
        SDMI_HIP(launch_fill_normal(bias.f(), cout, 13, stream_));
        ActQ q = new_actq(n, h, w, cin);
        SDMI_HIP(launch_quantize_fp8(x32.p, q.q, q.s, x32.rows(), cin, stream_));
        SDMI_HIP(launch_pack_conv_weight_fp8(w32.f(), bt8.p, bs8.p, cout, cin, 3, 3, stream_));
        ConvW cw; cw.cin = cin; cw.cout = cout; cw.k = 3; cw.dt = 1; cw.bias = bias.f(); cw.bt8 = bt8.f(); cw.bs8 = bs8.f();
        const int save_t = opt_fp8_tile_, save_s = opt_force_splits_;
        opt_fp8_tile_ = tile_cfg; opt_force_splits_ = splitk;
        float ms = 0;
        try {
            conv_fp8(cw, q, y, nullptr, nullptr);
            SDMI_HIP(hipEventRecord(ev0_, stream_));
            for (int i = 0; i < iters; ++i) conv_fp8(cw, q, y, nullptr, nullptr);
            SDMI_HIP(hipEventRecord(ev1_, stream_));
            SDMI_HIP(hipEventSynchronize(ev1_));
            SDMI_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
        } catch (...) {
            opt_fp8_tile_ = save_t; opt_force_splits_ = save_s;
            release(x32); release(y); release(q);
            throw;
        }
        opt_fp8_tile_ = save_t; opt_force_splits_ = save_s;
        release(x32); release(y); release(q);
        return (double)ms / std::max(1, iters);
    }
    const int wdt = (bf16_ && cin % 64 == 0) ? 1 : 0;
    Act a = new_act(n, h, w, cin, wdt), y = new_act(n, ho, wo, cout, (bf16_ && cout > 4) ? 1 : 0);
    Buf bt(this, (size_t)cout * cin * k * k * 4), bias(this, (size_t)cout * 4);
    {
        Buf tmp(this, std::max((size_t)a.rows() * cin, (size_t)cout * cin * k * k) * 4);
        SDMI_HIP(launch_fill_normal(wdt ? tmp.f() : a.p, (long long)a.rows() * cin, 11, stream_));
        if (wdt) SDMI_HIP(launch_f32_to_bf16(tmp.f(), a.p, (long long)a.rows() * cin, stream_));
        SDMI_HIP(launch_fill_normal(wdt ? tmp.f() : bt.f(), (long long)cout * cin * k * k, 12, stream_));
        if (wdt) SDMI_HIP(launch_f32_to_bf16(tmp.f(), bt.p, (long long)cout * cin * k * k, stream_));
        SDMI_HIP(hipStreamSynchronize(stream_));
    }
    SDMI_HIP(launch_fill_normal(bias.f(), cout, 13, stream_));
    TempSplit planes(this, bt.f(), wdt ? 0 : cout, (long long)cin * k * k);
    if (!wdt && tile_cfg >= 300 && cin % 32 == 0) {   // plane tiles are timed on planes their producer would have written
        a.p3 = pool_.alloc(a.bytes3()); a.ld3 = (cin / 32) * 192;
        SDMI_HIP(launch_split3_rows(a.p, a.p3, a.rows(), cin, cin, a.ld3, stream_));
    }
    ConvW cw; cw.cin = cin; cw.cout = cout; cw.k = k; cw.bt = bt.f(); cw.bias = bias.f(); cw.dt = wdt;
    const int save_t = opt_force_tile_, save_s = opt_force_splits_, save_p = opt_gemm_planes_;
    opt_force_tile_ = tile_cfg; opt_force_splits_ = splitk;
    if (tile_cfg >= 300 && !opt_gemm_planes_) opt_gemm_planes_ = 2;
    float ms = 0;
    try {
        conv(cw, a, y, stride, ups, nullptr, 0, nullptr);  // warm-up
        if (opt_bench_cold_) {
            // what the layer costs INSIDE the model: its weights come from HBM (5 GB of planes per UNet forward pass through the 256 MB
            // Infinity Cache between two uses), its activations from the L2s / Infinity Cache of the kernel that wrote them.  Between
            // timed launches a 512 MB fill evicts the weights, then the activation tensor is re-written (split3_rows / a copy).
            const size_t flush_bytes = (size_t)512 << 20;
            Buf flush(this, flush_bytes);
            Buf a_copy(this, a.bytes());
            SDMI_HIP(hipMemcpyAsync(a_copy.p, a.p, a.bytes(), hipMemcpyDeviceToDevice, stream_));
            for (int i = 0; i < iters; ++i) {
                SDMI_HIP(hipMemsetAsync(flush.p, i, flush_bytes, stream_));
                SDMI_HIP(hipMemcpyAsync(a.p, a_copy.p, a.bytes(), hipMemcpyDeviceToDevice, stream_));
                if (a.p3) SDMI_HIP(launch_split3_rows(a.p, a.p3, a.rows(), cin, cin, a.ld3, stream_));
                SDMI_HIP(hipEventRecord(ev0_, stream_));
                conv(cw, a, y, stride, ups, nullptr, 0, nullptr);
                SDMI_HIP(hipEventRecord(ev1_, stream_));
                SDMI_HIP(hipEventSynchronize(ev1_));
                float t = 0;
                SDMI_HIP(hipEventElapsedTime(&t, ev0_, ev1_));
                ms += t;
            }
        } else {
        SDMI_HIP(hipEventRecord(ev0_, stream_));
        for (int i = 0; i < iters; ++i) conv(cw, a, y, stride, ups, nullptr, 0, nullptr);
        SDMI_HIP(hipEventRecord(ev1_, stream_));
        SDMI_HIP(hipEventSynchronize(ev1_));
        SDMI_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
        }
        if (opt_gemm_probe_ && a.p3) {
            // diagnostic: one more launch in which every workgroup of the plane GEMM stamps its phases (ConvGemm::probe; 100 MHz clock)
            constexpr size_t kMaxBlocks = kGemmProbeBlocks;
            Buf pb(this, kMaxBlocks * 24 * sizeof(unsigned long long));
            SDMI_HIP(hipMemsetAsync(pb.p, 0, kMaxBlocks * 24 * sizeof(unsigned long long), stream_));
            if (opt_bench_cold_) {
                Buf flush(this, (size_t)512 << 20);
                SDMI_HIP(hipMemsetAsync(flush.p, 1, (size_t)512 << 20, stream_));
                SDMI_HIP(launch_split3_rows(a.p, a.p3, a.rows(), cin, cin, a.ld3, stream_));
            }
            probe_buf_ = static_cast<unsigned long long*>(pb.p);
            try { conv(cw, a, y, stride, ups, nullptr, 0, nullptr); } catch (...) { probe_buf_ = nullptr; throw; }
            probe_buf_ = nullptr;
            probe_report(pb.p, kMaxBlocks, n, cin, h, w, cout, k, tile_cfg, splitk);
        }
    } catch (...) {
        opt_force_tile_ = save_t; opt_force_splits_ = save_s; opt_gemm_planes_ = save_p;
        release(a); release(y);
        throw;
    }
    opt_force_tile_ = save_t; opt_force_splits_ = save_s; opt_gemm_planes_ = save_p;
    release(a); release(y);
    return (double)ms / std::max(1, iters);
}

double Engine::bench_attention(int n, int nq, int nk, int n_state, int n_head, int iters) {
    SDMI_HIP(hipSetDevice(cfg_.device));
    if (n <= 0 || nq <= 0 || nk <= 0 || n_head <= 0 || n_state % n_head) throw Error(SDMI_ERR_INVALID, "bench_attention: bad shape");
    const long long qe = (long long)n * nq * n_state, ke = (long long)n * nk * n_state;
    Buf q(this, qe * 4), k(this, ke * 4), v(this, ke * 4), o(this, qe * 4);
    SDMI_HIP(launch_fill_normal(q.f(), qe, 21, stream_));
    SDMI_HIP(launch_fill_normal(k.f(), ke, 22, stream_));
    SDMI_HIP(launch_fill_normal(v.f(), ke, 23, stream_));
    const int dt = edt();
    Buf qh(this, dt ? qe * 2 : 256), kh(this, dt ? ke * 2 : 256), vh(this, dt ? ke * 2 : 256);
    if (dt) {
        const int dh = n_state / n_head;
        SDMI_HIP(launch_f32_to_bf16_scaled(q.f(), qh.p, qe, q_prescaled(dt, dh) ? attn_bf16_q_scale(dh) : 1.f, stream_));
        SDMI_HIP(launch_f32_to_bf16(k.f(), kh.p, ke, stream_));
        SDMI_HIP(launch_f32_to_bf16(v.f(), vh.p, ke, stream_));
    }
    const float* qp = dt ? qh.f() : q.f(); const float* kp = dt ? kh.f() : k.f(); const float* vp = dt ? vh.f() : v.f();
    auto run = [&] {
        attention(qp, n_state, (long long)nq * n_state, kp, n_state, (long long)nk * n_state, vp, n_state, (long long)nk * n_state,
                  o.f(), n_state, (long long)nq * n_state, n, nq, nk, n_head, n_state / n_head, nullptr, nullptr, nullptr, 0, dt, nullptr, q_prescaled(dt, n_state / n_head));
    };
    run();
    float ms = 0;
    SDMI_HIP(hipEventRecord(ev0_, stream_));
    for (int i = 0; i < iters; ++i) run();
    SDMI_HIP(hipEventRecord(ev1_, stream_));
    SDMI_HIP(hipEventSynchronize(ev1_));
    SDMI_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
    return (double)ms / std::max(1, iters);
}

}  // namespace sdmi
