// multi.cpp -- the image batch sharded over the GPUs of one node, behind the C ABI.
//
// The reference's caller (src/bin/sample/main.rs:104-109) asks ONE StableDiffusion for n images of one prompt.  The
// path shards over independent images (GroupNorm, attention and DDIM are per sample), so the multi-GPU form is:
//   * one process, one Engine (weights replica, stream, activation pool) per device, one host thread per device;
//   * the prompt embedding lives on device 0 (where CLIP -- or the caller -- put it); ONE ncclBroadcast (RCCL over xGMI)
//     of the packed buffer [cond (T x ctx_dim) | uncond (Tu x ctx_dim)] = 473 088 B at T = Tu = 77 hands it to the other
//     devices, enqueued on each device's own stream inside one ncclGroupStart / ncclGroupEnd;
//   * device r samples the contiguous global image range [r*n/R, (r+1)*n/R) with noise keyed by the GLOBAL image index
//     (seed + i), decodes, and copies its u8 images into the caller's buffer: no other collective.
// RCCL is opened lazily (dlopen) when a multi-context is created: libsdmi.so itself depends on libamdhip64 only, so a
// process that already carries its own RCCL (PyTorch) and single-GPU users are unaffected.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"
#include "multi_ranks.hpp"

namespace sdmi {

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    void open() {
        if (lib) return;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) throw Error(SDMI_ERR_UNSUPPORTED, std::string("cannot load RCCL (librccl.so.1): ") + dlerror());
        auto sym = [&](const char* n) {
            void* p = dlsym(lib, n);
            if (!p) throw Error(SDMI_ERR_UNSUPPORTED, std::string("RCCL symbol missing: ") + n);
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    }
    void check(ncclResult_t r, const char* what) const {
        if (r != ncclSuccess) throw Error(SDMI_ERR_HIP, std::string(what) + ": " + (GetErrorString ? GetErrorString(r) : "RCCL error"));
    }
};
Rccl g_rccl;
}  // namespace

// contiguous image range of shard r of R over n images (the same rule as sharding.shard_range on the Python side)
void shard_range(int n, int r, int R, int* begin, int* end) {
    const int base = n / R, rem = n % R;
    *begin = r * base + (r < rem ? r : rem);
    *end = *begin + base + (r < rem ? 1 : 0);
}

class MultiEngine {
public:
    MultiEngine(const sdmi_config& cfg, const int* devices, int n) {
        if (!devices || n <= 0) throw Error(SDMI_ERR_INVALID, "create_multi: empty device list");
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) throw Error(SDMI_ERR_INVALID, "create_multi: a device is listed twice");
        devs_.assign(devices, devices + n);
        try {
            for (int i = 0; i < n; ++i) {
                sdmi_config c = cfg;
                c.device = devices[i];
                engines_.push_back(new Engine(c));
            }
            shards_.resize((size_t)n);
            g_rccl.open();
            comms_.resize(n, nullptr);
            g_rccl.check(g_rccl.CommInitAll(comms_.data(), n, devs_.data()), "ncclCommInitAll");
        } catch (...) {
            destroy();
            throw;
        }
    }
    ~MultiEngine() { destroy(); }
    MultiEngine(const MultiEngine&) = delete;
    MultiEngine& operator=(const MultiEngine&) = delete;

    int size() const { return (int)engines_.size(); }
    Engine& engine(int i) { return *engines_.at((size_t)i); }

    // runs f(rank) on one host thread per device and rethrows the first failure in the caller's thread (multi_ranks.hpp); on a failure
    // every device's stream is drained first, so nothing is still writing into the caller's buffers when the error reaches it
    template <class F>
    void for_each_device(F&& f) {
        run_on_ranks(size(), [&](int r) { return "device " + std::to_string(devs_[(size_t)r]); }, std::forward<F>(f),
                     [&] { for (int r = 0; r < size(); ++r) { (void)hipSetDevice(devs_[(size_t)r]); (void)hipStreamSynchronize(engine(r).stream()); } });
    }

    void sample_image(const float* context, int T, const float* uncond, int Tu, double scale, size_t n_steps, int n_images,
                      const float* init_latents, uint64_t seed, uint8_t* rgb_out) {
        if (!context || !uncond || !rgb_out) throw Error(SDMI_ERR_INVALID, "sample_image_sharded: null pointer");
        if (T <= 0 || Tu <= 0 || n_images <= 0) throw Error(SDMI_ERR_INVALID, "sample_image_sharded: T, Tu and n_images must be positive");
        const int R = size();
        const int cd = engine(0).config().ctx_dim, H = engine(0).latent_h(), W = engine(0).latent_w();
        const size_t n_cond = (size_t)T * cd, n_unc = (size_t)Tu * cd, n_prompt = n_cond + n_unc;
        const size_t lat_elems = (size_t)4 * H * W, img_bytes = (size_t)3 * 64 * H * W;

        // Per-device buffers live in the context and only ever grow (round 2 allocated four pool blocks per device per call and freed
        // them while the work that used them was still in flight); the prompt is staged through pinned host memory.
        for (int r = 0; r < R; ++r) {
            int g0, g1;
            shard_range(n_images, r, R, &g0, &g1);
            const size_t n = (size_t)(g1 - g0);
            SDMI_HIP(hipSetDevice(devs_[(size_t)r]));
            Shard& sh = shards_[(size_t)r];
            sh.prompt.reserve(n_prompt * sizeof(float));
            sh.ctx.reserve(n * n_cond * sizeof(float));
            sh.x0.reserve(n * lat_elems * sizeof(float));
            sh.lat.reserve(n * lat_elems * sizeof(float));
            sh.rgb.reserve(n * img_bytes);
        }
        SDMI_HIP(hipSetDevice(devs_[0]));
        if (pinned_bytes_ < n_prompt * sizeof(float)) {
            if (pinned_) (void)hipHostFree(pinned_);
            pinned_ = nullptr; pinned_bytes_ = 0;
            SDMI_HIP(hipHostMalloc(&pinned_, n_prompt * sizeof(float), hipHostMallocDefault));
            pinned_bytes_ = n_prompt * sizeof(float);
        }
        std::memcpy(pinned_, context, n_cond * sizeof(float));
        std::memcpy(reinterpret_cast<float*>(pinned_) + n_cond, uncond, n_unc * sizeof(float));
        SDMI_HIP(hipMemcpyAsync(shards_[0].prompt.p, pinned_, n_prompt * sizeof(float), hipMemcpyHostToDevice, engine(0).stream()));
        // THE collective of the path: one broadcast, each rank's part enqueued on that device's own stream
        g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
        for (int r = 0; r < R; ++r)
            g_rccl.check(g_rccl.Broadcast(shards_[(size_t)r].prompt.p, shards_[(size_t)r].prompt.p, n_prompt, ncclFloat32, 0, comms_[(size_t)r], engine(r).stream()), "ncclBroadcast");
        g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        ++broadcasts_;

        for_each_device([&](int r) {
            int g0, g1;
            shard_range(n_images, r, R, &g0, &g1);
            const int n = g1 - g0;
            Engine& e = engine(r);
            Shard& sh = shards_[(size_t)r];
            Engine::Call call(e);          // also hipSetDevice
            if (n > 0) {
                float* prompt = reinterpret_cast<float*>(sh.prompt.p);
                float* ctx = reinterpret_cast<float*>(sh.ctx.p);
                float* x0 = reinterpret_cast<float*>(sh.x0.p);
                float* lat = reinterpret_cast<float*>(sh.lat.p);
                // the same prompt for every image of the shard (sample/main.rs:100-109): one kernel, not n copies
                SDMI_HIP(launch_repeat_rows(prompt, ctx, n, (long long)n_cond, e.stream()));
                if (init_latents) {
                    SDMI_HIP(hipMemcpyAsync(x0, init_latents + (size_t)g0 * lat_elems, (size_t)n * lat_elems * sizeof(float), hipMemcpyHostToDevice, e.stream()));
                } else {
                    for (int i = 0; i < n; ++i)   // noise keyed by the global image index: independent of the device count
                        SDMI_HIP(launch_fill_normal(x0 + (size_t)i * lat_elems, (long long)lat_elems, seed + (uint64_t)(g0 + i), e.stream()));
                }
                e.sample_latent_dev(ctx, n, T, prompt + n_cond, Tu, scale, n_steps, x0, lat);
                e.decode_latent_dev(lat, n, (float)(1.0 / 0.18215), nullptr, reinterpret_cast<uint8_t*>(sh.rgb.p));
                SDMI_HIP(hipMemcpyAsync(rgb_out + (size_t)g0 * img_bytes, sh.rgb.p, (size_t)n * img_bytes, hipMemcpyDeviceToHost, e.stream()));
            }
            call.finish();   // waits for this device's stream (incl. its share of the broadcast and the D2H copy)
        });
    }

    long long broadcasts() const { return broadcasts_; }

private:
    // a device buffer that only grows (hipMalloc on the current device: the caller sets it)
    struct DevBuf {
        void* p = nullptr; size_t cap = 0;
        void reserve(size_t bytes) {
            if (bytes <= cap) return;
            if (p) { SDMI_HIP(hipDeviceSynchronize()); (void)hipFree(p); p = nullptr; cap = 0; }
            const size_t want = std::max<size_t>(bytes, 256);
            SDMI_HIP(hipMalloc(&p, want));
            cap = want;
        }
        void release() noexcept { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    };
    struct Shard { DevBuf prompt, ctx, x0, lat, rgb; };
    void destroy() noexcept {
        for (size_t i = 0; i < comms_.size(); ++i)
            if (comms_[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comms_[i]);
        comms_.clear();
        for (size_t r = 0; r < shards_.size() && r < devs_.size(); ++r) {
            (void)hipSetDevice(devs_[r]);
            (void)hipDeviceSynchronize();
            Shard& sh = shards_[r];
            sh.prompt.release(); sh.ctx.release(); sh.x0.release(); sh.lat.release(); sh.rgb.release();
        }
        shards_.clear();
        if (pinned_) (void)hipHostFree(pinned_);
        pinned_ = nullptr; pinned_bytes_ = 0;
        for (Engine* e : engines_) delete e;
        engines_.clear();
    }
    std::vector<Shard> shards_;
    void* pinned_ = nullptr;          // host staging of the packed prompt
    size_t pinned_bytes_ = 0;
    std::vector<int> devs_;
    std::vector<Engine*> engines_;
    std::vector<ncclComm_t> comms_;
    long long broadcasts_ = 0;
};

}  // namespace sdmi

// ---- C ABI (include/sdmi.h, "multi-GPU") ---------------------------------------------------------------------------
struct sdmi_multi {
    sdmi::MultiEngine* m;
    std::vector<sdmi_ctx> views;   // non-owning per-device handles for the single-device entry points (weights, options)
};

static void sdmi_set_last_error(const char* msg) { sdmi::set_last_error(msg); }

extern "C" {

static int multi_guard(const std::function<void()>& f) {
    try { f(); return SDMI_OK; }
    catch (const sdmi::Error& e) { sdmi_set_last_error(e.what()); return e.status; }
    catch (const std::exception& e) { sdmi_set_last_error(e.what()); return SDMI_ERR_INVALID; }
    catch (...) { sdmi_set_last_error("unknown internal error"); return SDMI_ERR_INVALID; }
}

int sdmi_create_multi(sdmi_multi** out, const sdmi_config* cfg, const int32_t* devices, int32_t n_devices) {
    if (!out || !cfg) { sdmi_set_last_error("sdmi_create_multi: null argument"); return SDMI_ERR_INVALID; }
    *out = nullptr;
    return multi_guard([&] {
        std::vector<int> devs(devices, devices + (devices && n_devices > 0 ? n_devices : 0));
        auto* mm = new sdmi_multi{new sdmi::MultiEngine(*cfg, devs.data(), (int)devs.size()), {}};
        for (int i = 0; i < mm->m->size(); ++i) mm->views.push_back(sdmi_ctx{&mm->m->engine(i)});
        *out = mm;
    });
}

void sdmi_destroy_multi(sdmi_multi* m) {
    if (!m) return;
    delete m->m;
    delete m;
}

int32_t sdmi_multi_size(sdmi_multi* m) { return m ? m->m->size() : SDMI_ERR_INVALID; }

sdmi_ctx* sdmi_multi_ctx(sdmi_multi* m, int32_t index) {
    if (!m || index < 0 || index >= m->m->size()) { sdmi_set_last_error("sdmi_multi_ctx: index out of range"); return nullptr; }
    return &m->views[(size_t)index];
}

int sdmi_multi_load_weights(sdmi_multi* m, const char* kind, const char* path) {
    return multi_guard([&] {
        if (!m || !kind || !path) throw sdmi::Error(SDMI_ERR_INVALID, "multi_load_weights: null argument");
        const std::string k = kind;
        if (k != "dump" && k != "burn") throw sdmi::Error(SDMI_ERR_INVALID, "multi_load_weights: kind must be \"dump\" or \"burn\"");
        m->m->for_each_device([&](int r) {
            sdmi::Engine& e = m->m->engine(r);
            if (k == "dump") e.load_weights_dir(path); else e.load_weights_mpk(path);
            e.finalize_weights();
        });
    });
}

int sdmi_sample_image_sharded(sdmi_multi* m, const float* context, int32_t T, const float* uncond, int32_t Tu, double scale,
                              size_t n_steps, int32_t n_images, const float* init_latents, uint64_t seed, uint8_t* rgb_out) {
    return multi_guard([&] {
        if (!m) throw sdmi::Error(SDMI_ERR_INVALID, "null sdmi_multi");
        m->m->sample_image(context, T, uncond, Tu, scale, n_steps, n_images, init_latents, seed, rgb_out);
    });
}

int sdmi_selftest_rank_errors(int32_t n_ranks, int32_t failing_rank) {
    return multi_guard([&] {
        if (n_ranks <= 0 || n_ranks > 64) throw sdmi::Error(SDMI_ERR_INVALID, "selftest_rank_errors: n_ranks out of range");
        std::vector<int> done((size_t)n_ranks, 0);
        int drained = 0;
        try {
            sdmi::run_on_ranks(n_ranks, [](int r) { return "rank " + std::to_string(r); },
                               [&](int r) {
                                   if (r == failing_rank) throw sdmi::Error(SDMI_ERR_HIP, "injected failure");
                                   done[(size_t)r] = 1;
                               },
                               [&] { ++drained; });
        } catch (const sdmi::Error& e) {
            // every other rank must have run to completion and the drain hook exactly once before the error surfaces
            for (int r = 0; r < n_ranks; ++r)
                if (r != failing_rank && !done[(size_t)r]) throw sdmi::Error(SDMI_ERR_STATE, "selftest: a healthy rank was abandoned");
            if (drained != 1) throw sdmi::Error(SDMI_ERR_STATE, "selftest: the drain hook did not run exactly once");
            throw;
        }
        if (failing_rank >= 0 && failing_rank < n_ranks) throw sdmi::Error(SDMI_ERR_STATE, "selftest: the failure was swallowed");
    });
}

int64_t sdmi_multi_broadcast_count(sdmi_multi* m) { return m ? m->m->broadcasts() : SDMI_ERR_INVALID; }

int sdmi_shard_range(int32_t n_images, int32_t rank, int32_t n_ranks, int32_t* begin, int32_t* end) {
    if (!begin || !end || n_images < 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) { sdmi_set_last_error("sdmi_shard_range: bad argument"); return SDMI_ERR_INVALID; }
    int b = 0, e = 0;
    sdmi::shard_range(n_images, rank, n_ranks, &b, &e);
    *begin = b; *end = e;
    return SDMI_OK;
}

}  // extern "C"
