// sdmi_capi.cpp -- extern "C" boundary of libsdmi.so (see include/sdmi.h).
// Translates C arguments to Engine calls, C++ exceptions to sdmi_status codes,
// and stages host buffers through the device pool for the host-pointer API.
#include <cstring>
#include <string>

#include "engine.hpp"
#include "mpk_reader.hpp"
#include "tokenizer.hpp"

using sdmi::Engine;
using sdmi::Error;

namespace sdmi {
void write_png_rgb8(const std::string& path, const uint8_t* rgb, int width, int height);  // png_writer.cpp
}

struct sdmi_tokenizer {
    sdmi::Tokenizer tok;
};

static thread_local std::string g_last_error;

namespace sdmi {
void set_last_error(const std::string& msg) { g_last_error = msg; }
}

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return SDMI_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_last_error = std::string("internal error: ") + e.what();
        return SDMI_ERR_INVALID;
    } catch (...) {
        g_last_error = "unknown internal error";
        return SDMI_ERR_INVALID;
    }
}

static Engine& eng(sdmi_ctx* c) {
    if (!c || !c->engine) throw Error(SDMI_ERR_INVALID, "null sdmi_ctx");
    return *c->engine;
}

namespace {
// host <-> device staging on the context stream
struct DevIn {
    Engine::Buf buf;
    DevIn(Engine& e, const void* host, size_t bytes) : buf(&e, bytes) {
        if (!host) throw Error(SDMI_ERR_INVALID, "null input pointer");
        SDMI_HIP(hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, e.stream()));
    }
    const float* f() const { return buf.f(); }
};
struct DevOut {
    Engine& e; Engine::Buf buf; void* host; size_t bytes;
    DevOut(Engine& e_, void* host_, size_t bytes_) : e(e_), buf(&e_, bytes_), host(host_), bytes(bytes_) {
        if (!host) throw Error(SDMI_ERR_INVALID, "null output pointer");
    }
    void fetch() {
        SDMI_HIP(hipMemcpyAsync(host, buf.p, bytes, hipMemcpyDeviceToHost, e.stream()));
        SDMI_HIP(hipStreamSynchronize(e.stream()));
    }
    float* f() const { return buf.f(); }
};
}  // namespace

extern "C" {

int sdmi_default_config(sdmi_config* cfg) {
    if (!cfg) return SDMI_ERR_INVALID;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->model_channels = 320;  // unet/mod.rs:41
    cfg->n_head = 8;            // unet/mod.rs:44
    cfg->ctx_dim = 768;         // unet/mod.rs:44
    cfg->latent_h = 64;         // stablediffusion/mod.rs:116
    cfg->latent_w = 64;
    cfg->vae_ch = 128;          // autoencoder/mod.rs:33-34
    cfg->max_batch = 0;
    cfg->precision = 0;
    cfg->clip_layers = 12;      // CLIPConfig::new(49408, 768, 12, 77, 12), stablediffusion/mod.rs:29
    cfg->clip_heads = 12;
    cfg->clip_vocab = 49408;
    cfg->clip_ctx = 77;
    return SDMI_OK;
}

const char* sdmi_version(void) { return "sdmi 0.3 gfx950 fp32+bf16+mxfp8 (MI355X-native SD v1.4: CLIP, UNet DDIM/CFG loop, VAE; multi-GPU sharding)"; }

const char* sdmi_last_error(void) { return g_last_error.c_str(); }

int sdmi_create(sdmi_ctx** out, const sdmi_config* cfg) {
    if (!out || !cfg) { g_last_error = "sdmi_create: null argument"; return SDMI_ERR_INVALID; }
    *out = nullptr;
    return guarded([&] {
        Engine* e = new Engine(*cfg);
        *out = new sdmi_ctx{e};
    });
}

void sdmi_destroy(sdmi_ctx* ctx) {
    if (!ctx) return;
    delete ctx->engine;
    delete ctx;
}

int sdmi_synchronize(sdmi_ctx* ctx) { return guarded([&] { eng(ctx).sync(); }); }

int sdmi_set_stream(sdmi_ctx* ctx, void* hip_stream, int32_t enable) {
    return guarded([&] { eng(ctx).set_user_stream(reinterpret_cast<hipStream_t>(hip_stream), enable != 0); });
}

int sdmi_set_weight(sdmi_ctx* ctx, const char* name, const float* data, int32_t ndim, const int64_t* dims) {
    return guarded([&] { eng(ctx).set_weight(name, data, ndim, dims); });
}

int sdmi_weight_count(sdmi_ctx* ctx) {
    int n = 0;
    int st = guarded([&] { n = (int)eng(ctx).entries().size(); });
    return st == SDMI_OK ? n : st;
}

int sdmi_weight_info(sdmi_ctx* ctx, int32_t index, const char** name, int32_t* ndim, int64_t dims[4]) {
    return guarded([&] {
        const auto& es = eng(ctx).entries();
        if (index < 0 || index >= (int)es.size()) throw Error(SDMI_ERR_INVALID, "weight index out of range");
        if (name) *name = es[index].name.c_str();
        if (ndim) *ndim = es[index].ndim;
        if (dims) for (int i = 0; i < 4; ++i) dims[i] = es[index].dims[i];
    });
}

int sdmi_load_weights_dir(sdmi_ctx* ctx, const char* dump_dir) {
    return guarded([&] { eng(ctx).load_weights_dir(dump_dir); });
}

int sdmi_load_weights_mpk(sdmi_ctx* ctx, const char* mpk_path) {
    return guarded([&] { eng(ctx).load_weights_mpk(mpk_path); });
}

int sdmi_mpk_list(const char* mpk_path, char* out, size_t capacity, size_t* needed) {
    return guarded([&] {
        if (!mpk_path || !needed) throw Error(SDMI_ERR_INVALID, "mpk_list: null argument");
        sdmi::MpkFile f(mpk_path);
        std::string s = "# format=" + f.format() + " float=" + f.float_type() + "\n";
        for (const auto& t : f.tensors()) {
            s += t.name + "\t";
            for (size_t i = 0; i < t.shape.size(); ++i) s += (i ? "," : "") + std::to_string(t.shape[i]);
            s += "\t" + std::to_string(t.file_offset) + "\n";
        }
        *needed = s.size() + 1;
        if (out && capacity >= s.size() + 1) std::memcpy(out, s.c_str(), s.size() + 1);
        else if (out && capacity) throw Error(SDMI_ERR_INVALID, "mpk_list: capacity too small");
    });
}

int sdmi_load_weights_packed(sdmi_ctx* ctx, const float* data, size_t n_floats, int32_t groups) {
    return guarded([&] { eng(ctx).load_weights_packed(data, n_floats, groups); });
}

int64_t sdmi_packed_size(sdmi_ctx* ctx, int32_t groups) {
    int64_t n = 0;
    int st = guarded([&] { n = (int64_t)eng(ctx).packed_size(groups); });
    return st == SDMI_OK ? n : st;
}

int sdmi_finalize_weights(sdmi_ctx* ctx) { return guarded([&] { eng(ctx).finalize_weights(); }); }

// ---- hot path, device pointers -----------------------------------------------------
int sdmi_sample_latent_dev(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T, const float* uncond, int32_t Tu,
                           double scale, size_t n_steps, const float* init_latent, float* latent_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (!context || !uncond || !init_latent || !latent_out) throw Error(SDMI_ERR_INVALID, "sample_latent_dev: null pointer");
        Engine::Call call(e, /*dev_inputs=*/true);
        e.sample_latent_dev(context, n, T, uncond, Tu, scale, n_steps, init_latent, latent_out);
        call.finish();
    });
}

int sdmi_latent_to_image_dev(sdmi_ctx* ctx, const float* latent, int32_t n, uint8_t* rgb_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (!latent || !rgb_out) throw Error(SDMI_ERR_INVALID, "latent_to_image_dev: null pointer");
        Engine::Call call(e, /*dev_inputs=*/true);
        e.decode_latent_dev(latent, n, (float)(1.0 / 0.18215), nullptr, rgb_out);
        call.finish();
    });
}

int sdmi_sample_image_dev(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T, const float* uncond, int32_t Tu,
                          double scale, size_t n_steps, const float* init_latent, uint8_t* rgb_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (!context || !uncond || !init_latent || !rgb_out) throw Error(SDMI_ERR_INVALID, "sample_image_dev: null pointer");
        if (n <= 0) throw Error(SDMI_ERR_INVALID, "sample_image_dev: n must be positive");
        Engine::Call call(e, /*dev_inputs=*/true);
        Engine::Buf lat(&e, (size_t)n * 4 * e.latent_h() * e.latent_w() * sizeof(float));
        e.sample_latent_dev(context, n, T, uncond, Tu, scale, n_steps, init_latent, lat.f());
        e.decode_latent_dev(lat.f(), n, (float)(1.0 / 0.18215), nullptr, rgb_out);
        call.finish();
    });
}

// ---- hot path, host pointers ---------------------------------------------------------
int sdmi_unet_forward(sdmi_ctx* ctx, const float* x, int32_t t, const float* context, int32_t n, int32_t T, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || T <= 0) throw Error(SDMI_ERR_INVALID, "unet_forward: n and T must be positive");
        const size_t lat = (size_t)n * 4 * e.latent_h() * e.latent_w() * sizeof(float);
        Engine::Call call(e);
        DevIn dx(e, x, lat), dc(e, context, (size_t)n * T * e.config().ctx_dim * sizeof(float));
        DevOut dout(e, out, lat);
        e.unet_forward_dev(dx.f(), t, dc.f(), n, T, dout.f());
        call.finish();
        dout.fetch();
    });
}

// ---- tokenizer + CLIP -----------------------------------------------------------------
int sdmi_tokenizer_create(sdmi_tokenizer** out, const char* merges_path) {
    if (!out || !merges_path) { g_last_error = "sdmi_tokenizer_create: null argument"; return SDMI_ERR_INVALID; }
    *out = nullptr;
    return guarded([&] { *out = new sdmi_tokenizer{sdmi::Tokenizer(merges_path)}; });
}

void sdmi_tokenizer_destroy(sdmi_tokenizer* tok) { delete tok; }

int sdmi_tokenizer_vocab_size(const sdmi_tokenizer* tok) { return tok ? tok->tok.vocab_size() : SDMI_ERR_INVALID; }

int sdmi_tokenizer_encode(const sdmi_tokenizer* tok, const char* text, int32_t* ids, int32_t capacity, int32_t* n_ids) {
    return guarded([&] {
        if (!tok || !text || !n_ids) throw Error(SDMI_ERR_INVALID, "tokenizer_encode: null argument");
        const std::vector<int32_t> v = tok->tok.encode(text);
        *n_ids = (int32_t)v.size();
        if ((int64_t)v.size() > capacity) throw Error(SDMI_ERR_INVALID, "tokenizer_encode: capacity too small");
        if (!v.empty() && !ids) throw Error(SDMI_ERR_INVALID, "tokenizer_encode: null output");
        std::copy(v.begin(), v.end(), ids);
    });
}

int sdmi_tokenizer_decode(const sdmi_tokenizer* tok, const int32_t* ids, int32_t n, char* out, int32_t capacity, int32_t* n_bytes) {
    return guarded([&] {
        if (!tok || !n_bytes || n < 0 || (n > 0 && !ids)) throw Error(SDMI_ERR_INVALID, "tokenizer_decode: bad argument");
        const std::string s = tok->tok.decode(ids, (size_t)n);
        *n_bytes = (int32_t)s.size();
        if ((int64_t)s.size() > capacity) throw Error(SDMI_ERR_INVALID, "tokenizer_decode: capacity too small");
        if (!s.empty() && !out) throw Error(SDMI_ERR_INVALID, "tokenizer_decode: null output");
        std::memcpy(out, s.data(), s.size());
    });
}

static void clip_forward_host(Engine& e, const int32_t* tokens, int n, int T, float* out) {
    if (!tokens || !out) throw Error(SDMI_ERR_INVALID, "clip_forward: null argument");
    if (n <= 0 || T <= 0) throw Error(SDMI_ERR_INVALID, "clip_forward: n and seq_len must be positive");
    const int vocab = e.config().clip_vocab;
    for (long long i = 0; i < (long long)n * T; ++i)   // the reference's embedding gather panics on an id outside the table
        if (tokens[i] < 0 || tokens[i] >= vocab) throw Error(SDMI_ERR_INVALID, "clip_forward: token id outside the vocabulary");
    Engine::Call call(e);
    DevIn dt(e, tokens, (size_t)n * T * sizeof(int32_t));
    DevOut dout(e, out, (size_t)n * T * e.config().ctx_dim * sizeof(float));
    e.clip_forward_dev(reinterpret_cast<const int32_t*>(dt.buf.p), n, T, dout.f());
    call.finish();
    dout.fetch();
}

int sdmi_clip_forward(sdmi_ctx* ctx, const int32_t* tokens, int32_t n, int32_t seq_len, float* out) {
    return guarded([&] { clip_forward_host(eng(ctx), tokens, n, seq_len, out); });
}

int sdmi_context(sdmi_ctx* ctx, const sdmi_tokenizer* tok, const char* text, float* out, int32_t capacity_tokens, int32_t* T) {
    return guarded([&] {
        if (!tok || !text || !T) throw Error(SDMI_ERR_INVALID, "context: null argument");
        Engine& e = eng(ctx);
        const std::vector<int32_t> ids = tok->tok.encode(std::string("<|startoftext|>") + text + "<|endoftext|>");  // mod.rs:200
        *T = (int32_t)ids.size();
        if ((int64_t)ids.size() > capacity_tokens) throw Error(SDMI_ERR_INVALID, "context: capacity_tokens too small");
        clip_forward_host(e, ids.data(), 1, (int)ids.size(), out);
    });
}

int sdmi_encode_image(sdmi_ctx* ctx, const float* img, int32_t n, float* latent_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0) throw Error(SDMI_ERR_INVALID, "encode_image: n must be positive");
        const size_t lat = (size_t)n * 4 * e.latent_h() * e.latent_w() * sizeof(float);
        Engine::Call call(e);
        DevIn di(e, img, lat * 48);   // 3 * 64 / 4
        DevOut dout(e, latent_out, lat);
        e.encode_image_dev(di.f(), n, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_write_png(const char* path, const uint8_t* rgb, int32_t width, int32_t height) {
    return guarded([&] {
        if (!path) throw Error(SDMI_ERR_INVALID, "write_png: null path");
        sdmi::write_png_rgb8(path, rgb, width, height);
    });
}

static void make_init_latent(Engine& e, const float* init_latent, uint64_t seed, int n, Engine::Buf& dst) {
    const size_t per = (size_t)4 * e.latent_h() * e.latent_w();
    if (init_latent) {
        SDMI_HIP(hipMemcpyAsync(dst.p, init_latent, n * per * sizeof(float), hipMemcpyHostToDevice, e.stream()));
    } else {
        for (int i = 0; i < n; ++i) SDMI_HIP(sdmi::launch_fill_normal(dst.f() + i * per, (long long)per, seed + (uint64_t)i, e.stream()));
    }
}

int sdmi_sample_latent(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T, const float* uncond, int32_t Tu,
                       double scale, size_t n_steps, const float* init_latent, uint64_t seed, float* latent_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || T <= 0 || Tu <= 0) throw Error(SDMI_ERR_INVALID, "sample_latent: n, T, Tu must be positive");
        const int cd = e.config().ctx_dim;
        const size_t lat = (size_t)n * 4 * e.latent_h() * e.latent_w() * sizeof(float);
        Engine::Call call(e);
        DevIn dc(e, context, (size_t)n * T * cd * sizeof(float)), du(e, uncond, (size_t)Tu * cd * sizeof(float));
        Engine::Buf x0(&e, lat);
        make_init_latent(e, init_latent, seed, n, x0);
        DevOut dout(e, latent_out, lat);
        e.sample_latent_dev(dc.f(), n, T, du.f(), Tu, scale, n_steps, x0.f(), dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_decode_latent(sdmi_ctx* ctx, const float* latent, int32_t n, float* img_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0) throw Error(SDMI_ERR_INVALID, "decode_latent: n must be positive");
        const size_t hw = (size_t)e.latent_h() * e.latent_w();
        Engine::Call call(e);
        DevIn dl(e, latent, (size_t)n * 4 * hw * sizeof(float));
        DevOut dout(e, img_out, (size_t)n * 3 * 64 * hw * sizeof(float));
        e.decode_latent_dev(dl.f(), n, 1.0f, dout.f(), nullptr);
        call.finish();
        dout.fetch();
    });
}

int sdmi_latent_to_image(sdmi_ctx* ctx, const float* latent, int32_t n, uint8_t* rgb_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0) throw Error(SDMI_ERR_INVALID, "latent_to_image: n must be positive");
        const size_t hw = (size_t)e.latent_h() * e.latent_w();
        Engine::Call call(e);
        DevIn dl(e, latent, (size_t)n * 4 * hw * sizeof(float));
        DevOut dout(e, rgb_out, (size_t)n * 3 * 64 * hw);
        e.decode_latent_dev(dl.f(), n, (float)(1.0 / 0.18215), nullptr, reinterpret_cast<uint8_t*>(dout.buf.p));
        call.finish();
        dout.fetch();
    });
}

int sdmi_sample_image(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T, const float* uncond, int32_t Tu,
                      double scale, size_t n_steps, const float* init_latent, uint64_t seed, uint8_t* rgb_out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || T <= 0 || Tu <= 0) throw Error(SDMI_ERR_INVALID, "sample_image: n, T, Tu must be positive");
        const int cd = e.config().ctx_dim;
        const size_t hw = (size_t)e.latent_h() * e.latent_w();
        const size_t lat = (size_t)n * 4 * hw * sizeof(float);
        Engine::Call call(e);
        DevIn dc(e, context, (size_t)n * T * cd * sizeof(float)), du(e, uncond, (size_t)Tu * cd * sizeof(float));
        Engine::Buf x0(&e, lat), xl(&e, lat);
        make_init_latent(e, init_latent, seed, n, x0);
        DevOut dout(e, rgb_out, (size_t)n * 3 * 64 * hw);
        e.sample_latent_dev(dc.f(), n, T, du.f(), Tu, scale, n_steps, x0.f(), xl.f());
        e.decode_latent_dev(xl.f(), n, (float)(1.0 / 0.18215), nullptr, reinterpret_cast<uint8_t*>(dout.buf.p));
        call.finish();
        dout.fetch();
    });
}

int sdmi_qkv_attention(sdmi_ctx* ctx, const float* q, const float* k, const float* v, const float* mask,
                       int32_t mask_ld, int32_t n, int32_t nq, int32_t nk, int32_t n_state, int32_t n_head, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || nq <= 0 || nk <= 0 || n_state <= 0) throw Error(SDMI_ERR_INVALID, "qkv_attention: bad shape");
        const size_t qb = (size_t)n * nq * n_state * sizeof(float), kb = (size_t)n * nk * n_state * sizeof(float);
        Engine::Call call(e);
        DevIn dq(e, q, qb), dk(e, k, kb), dv(e, v, kb);
        Engine::Buf dm(&e, mask ? (size_t)nq * mask_ld * sizeof(float) : 256);
        if (mask) SDMI_HIP(hipMemcpyAsync(dm.p, mask, (size_t)nq * mask_ld * sizeof(float), hipMemcpyHostToDevice, e.stream()));
        DevOut dout(e, out, qb);
        e.qkv_attention_dev(dq.f(), dk.f(), dv.f(), mask ? dm.f() : nullptr, mask_ld, n, nq, nk, n_state, n_head, dout.f());
        call.finish();
        dout.fetch();
    });
}

// ---- operator-level entry points -----------------------------------------------------------
int sdmi_op_group_norm(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta, int32_t n, int32_t c,
                       int32_t h, int32_t w, int32_t n_group, float eps, int32_t fuse_silu, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || c <= 0 || h <= 0 || w <= 0) throw Error(SDMI_ERR_INVALID, "group_norm: bad shape");
        const size_t bytes = (size_t)n * c * h * w * sizeof(float);
        Engine::Call call(e);
        DevIn dx(e, x, bytes), dg(e, gamma, c * sizeof(float)), db(e, beta, c * sizeof(float));
        DevOut dout(e, out, bytes);
        e.op_group_norm(dx.f(), dg.f(), db.f(), n, c, h, w, n_group, eps, fuse_silu != 0, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_group_norm_fp8(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta, int32_t n, int32_t c, int32_t h,
                           int32_t w, int32_t n_group, float eps, int32_t fuse_silu, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || c <= 0 || h <= 0 || w <= 0) throw Error(SDMI_ERR_INVALID, "group_norm_fp8: bad shape");
        const size_t bytes = (size_t)n * c * h * w * sizeof(float);
        Engine::Call call(e);
        DevIn dx(e, x, bytes), dg(e, gamma, c * sizeof(float)), db(e, beta, c * sizeof(float));
        DevOut dout(e, out, bytes);
        e.op_group_norm_fp8(dx.f(), dg.f(), db.f(), n, c, h, w, n_group, eps, fuse_silu != 0, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_layer_norm(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta, int32_t rows, int32_t c,
                       float eps, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (rows <= 0 || c <= 0) throw Error(SDMI_ERR_INVALID, "layer_norm: bad shape");
        const size_t bytes = (size_t)rows * c * sizeof(float);
        Engine::Call call(e);
        DevIn dx(e, x, bytes), dg(e, gamma, c * sizeof(float)), db(e, beta, c * sizeof(float));
        DevOut dout(e, out, bytes);
        e.op_layer_norm(dx.f(), dg.f(), db.f(), rows, c, eps, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_conv2d(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias, int32_t n, int32_t cin,
                   int32_t h, int32_t w, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t upsample2x,
                   float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || cout <= 0 || stride <= 0) throw Error(SDMI_ERR_INVALID, "conv2d: bad shape");
        const int ups = upsample2x ? 1 : 0;
        const int hin = h << ups, win = w << ups;
        const int ho = (hin + 2 * pad - k) / stride + 1, wo = (win + 2 * pad - k) / stride + 1;
        Engine::Call call(e);
        DevIn dx(e, x, (size_t)n * cin * h * w * sizeof(float)), dw(e, weight, (size_t)cout * cin * k * k * sizeof(float));
        Engine::Buf db(&e, (size_t)cout * sizeof(float));
        if (bias) SDMI_HIP(hipMemcpyAsync(db.p, bias, (size_t)cout * sizeof(float), hipMemcpyHostToDevice, e.stream()));
        DevOut dout(e, out, (size_t)n * cout * ho * wo * sizeof(float));
        e.op_conv2d(dx.f(), dw.f(), bias ? db.f() : nullptr, n, cin, h, w, cout, k, stride, pad, ups, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_geglu_forward(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias, int32_t rows, int32_t cin,
                          int32_t hidden, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (rows <= 0 || cin <= 0 || hidden <= 0 || cin % 32) throw Error(SDMI_ERR_INVALID, "geglu_forward: bad shape");
        Engine::Call call(e);
        DevIn dx(e, x, (size_t)rows * cin * sizeof(float)), dw(e, weight, (size_t)cin * 2 * hidden * sizeof(float));
        Engine::Buf db(&e, (size_t)2 * hidden * sizeof(float));
        if (bias) SDMI_HIP(hipMemcpyAsync(db.p, bias, (size_t)2 * hidden * sizeof(float), hipMemcpyHostToDevice, e.stream()));
        DevOut dout(e, out, (size_t)rows * hidden * sizeof(float));
        e.op_geglu_forward(dx.f(), dw.f(), bias ? db.f() : nullptr, rows, cin, hidden, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_linear(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias, int32_t rows, int32_t cin,
                   int32_t cout, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (rows <= 0 || cin <= 0 || cout <= 0) throw Error(SDMI_ERR_INVALID, "linear: bad shape");
        Engine::Call call(e);
        DevIn dx(e, x, (size_t)rows * cin * sizeof(float)), dw(e, weight, (size_t)cin * cout * sizeof(float));
        Engine::Buf db(&e, (size_t)cout * sizeof(float));
        if (bias) SDMI_HIP(hipMemcpyAsync(db.p, bias, (size_t)cout * sizeof(float), hipMemcpyHostToDevice, e.stream()));
        DevOut dout(e, out, (size_t)rows * cout * sizeof(float));
        e.op_linear(dx.f(), dw.f(), bias ? db.f() : nullptr, rows, cin, cout, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_geglu(sdmi_ctx* ctx, const float* proj, int32_t rows, int32_t hidden, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (rows <= 0 || hidden <= 0 || hidden % 4) throw Error(SDMI_ERR_INVALID, "geglu: bad shape");
        Engine::Call call(e);
        DevIn dp(e, proj, (size_t)rows * 2 * hidden * sizeof(float));
        DevOut dout(e, out, (size_t)rows * hidden * sizeof(float));
        e.op_geglu(dp.f(), rows, hidden, dout.f());
        call.finish();
        dout.fetch();
    });
}

int sdmi_op_timestep_embedding(sdmi_ctx* ctx, int32_t t, int32_t dim, float* out) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (dim <= 0 || dim % 2) throw Error(SDMI_ERR_INVALID, "timestep_embedding: dim must be even");
        Engine::Call call(e);
        DevOut dout(e, out, (size_t)dim * sizeof(float));
        e.op_timestep_embedding(t, dim, dout.f());
        call.finish();
        dout.fetch();
    });
}

// ---- tuning / introspection --------------------------------------------------------------------
int sdmi_set_option(sdmi_ctx* ctx, const char* key, const char* value) {
    return guarded([&] {
        if (!key || !value) throw Error(SDMI_ERR_INVALID, "set_option: null argument");
        eng(ctx).set_option(key, value);
    });
}

int sdmi_last_call_stats(sdmi_ctx* ctx, double* gpu_ms, int64_t* n_kernels, double* flops) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (gpu_ms) *gpu_ms = e.last_ms;
        if (n_kernels) *n_kernels = e.last_kernels;
        if (flops) *flops = e.last_flops;
    });
}

int sdmi_profile_stats(sdmi_ctx* ctx, int32_t cls, double* ms, int64_t* launches, double* flops, double* bytes) {
    return guarded([&] {
        Engine& e = eng(ctx);
        if (cls < 0 || cls >= Engine::PC_COUNT) throw Error(SDMI_ERR_INVALID, "profile_stats: class out of range");
        e.prof_flush();
        if (ms) *ms = e.prof_[cls].ms;
        if (launches) *launches = e.prof_[cls].launches;
        if (flops) *flops = e.prof_[cls].flops;
        if (bytes) *bytes = e.prof_[cls].bytes;
    });
}

int sdmi_profile_overhead(sdmi_ctx* ctx, double* ms) {
    return guarded([&] {
        if (!ms) throw Error(SDMI_ERR_INVALID, "profile_overhead: null output");
        *ms = eng(ctx).prof_overhead_ms_;
    });
}

int sdmi_bench_conv(sdmi_ctx* ctx, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t k,
                    int32_t stride, int32_t upsample2x, int32_t tile_cfg, int32_t splitk, int32_t iters, double* ms_out) {
    return guarded([&] {
        if (!ms_out) throw Error(SDMI_ERR_INVALID, "bench_conv: null output");
        *ms_out = eng(ctx).bench_conv(n, cin, h, w, cout, k, stride, upsample2x ? 1 : 0, tile_cfg, splitk, iters);
    });
}

int sdmi_bench_attention(sdmi_ctx* ctx, int32_t n, int32_t nq, int32_t nk, int32_t n_state, int32_t n_head,
                         int32_t iters, double* ms_out) {
    return guarded([&] {
        if (!ms_out) throw Error(SDMI_ERR_INVALID, "bench_attention: null output");
        *ms_out = eng(ctx).bench_attention(n, nq, nk, n_state, n_head, iters);
    });
}

}  // extern "C"
