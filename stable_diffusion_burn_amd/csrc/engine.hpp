// engine.hpp -- host runtime of libsdmi: device pool, weight registry, the static
// UNet / VAE-decoder launch graphs and the DDIM+CFG sampler.
//
// Mirrors the reference's L4/L3 structure:
//   Engine::sample_latent   <- StableDiffusion::sample_latent  (stablediffusion/mod.rs:102-160)
//   Engine::unet_run        <- UNet::forward                   (unet/mod.rs:109-143)
//   Engine::decode          <- Autoencoder::decode_latent      (autoencoder/mod.rs:68-71,205-217)
// but is not a translation: activations are NHWC, weights are pre-packed for the
// implicit-GEMM kernel, cond+uncond run as ONE batch-2n forward, time-embedding
// projections and cross-attention K/V are hoisted out of the step loop, and all
// launches go to one HIP stream with no host synchronisation inside the loop.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sdmi.h"
#include "error.hpp"
#include "kernels.hpp"

namespace sdmi {

#define SDMI_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw ::sdmi::Error(SDMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + \
                                                  __FILE__ + ":" + std::to_string(__LINE__) + ")");      \
    } while (0)

// ---- device memory pool: host-side first-fit allocator over big slabs -------------
// All work is on one stream, so a block may be reused as soon as it is freed
// (kernel order == program order).  Addresses are deterministic for a given call
// sequence, which keeps launches graph-capturable.
class DevPool {
public:
    ~DevPool();
    void* alloc(size_t bytes);
    void free(void* p);
    size_t reserved() const { return reserved_; }
    size_t high_water() const { return high_; }
    // blocks are numbered in allocation order; free_since(m) returns every block allocated after mark m = serial()
    // (the clean-up of a call that threw half way through a forward pass)
    unsigned long long serial() const { return serial_; }
    size_t free_since(unsigned long long mark);

private:
    struct Block { size_t off, size; };
    struct Slab { char* base; size_t size; std::vector<Block> free_list; };
    std::vector<Slab> slabs_;
    struct Live { int slab; size_t size; unsigned long long serial; };
    std::map<void*, Live> live_;
    size_t reserved_ = 0, in_use_ = 0, high_ = 0;
    unsigned long long serial_ = 0;
};

// dt: storage type of a device tensor -- 0 = fp32, 1 = bf16 (precision = 1).  `p` is typed float* for
// historical reasons; for dt == 1 it is an opaque pointer to 2-byte elements.
// ld: elements between consecutive pixels (0 = dense, i.e. c).  A channel-slice VIEW of a wider buffer (view = true, never
// freed) is how Tensor::cat (unet/mod.rs:134) is realised without a copy: producers write their slice, the consumer reads the whole.
// p3 (precision = 0 only): the same tensor as three bf16 planes [pixel][ld3 / 192 slices][h, m, l][32] (k_split3.hpp), what k_gemm3p.hip
// reads; written by the tensor's producer.  A tensor may exist as fp32 (p), as planes (p3), or as both.
struct Act {
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    int dt = 0;
    int ld = 0;
    bool view = false;
    void* p3 = nullptr;
    int ld3 = 0;       // bytes between pixels of p3
    size_t bytes3() const { return (size_t)rows() * (size_t)(c / 32) * 192; }
    long long rows() const { return (long long)n * h * w; }
    int stride() const { return ld ? ld : c; }
    size_t bytes() const { return (size_t)rows() * c * (dt ? 2 : 4); }
};

// bt8 / bs8 (precision = 2 only): the same weight as MXFP8 -- e4m3 [cout][Kp] + E8M0 scales [cout][Kp / 32] (k_fp8.hip)
struct ConvW { float* bt = nullptr; float* bias = nullptr; int cin = 0, cout = 0, k = 1; int dt = 0; float* bt8 = nullptr; float* bs8 = nullptr; };
// an MXFP8 activation: e4m3 [n][h][w][cp] + E8M0 scales [n][h][w][cp / 32], cp = c rounded up to 128 (zero padded)
struct ActQ {
    void* q = nullptr; void* s = nullptr;
    int n = 0, h = 0, w = 0, c = 0, cp = 0;
    long long rows() const { return (long long)n * h * w; }
};
struct LinW { float* bt = nullptr; float* bias = nullptr; int cin = 0, cout = 0; int dt = 0; float* bt8 = nullptr; float* bs8 = nullptr; };   // bt8 / bs8: MXFP8 copy (precision = 2, option fp8_linear)
struct NormW { float* gamma = nullptr; float* beta = nullptr; int c = 0; float eps = 1e-5f; };  // eps: Q3 default, overridden by the dump's eps file

struct ResW {  // UNet ResBlock (unet/mod.rs:700-734) and VAE ResnetBlock (autoencoder/mod.rs:503-528)
    NormW norm_in; ConvW conv_in; LinW lin_embed; NormW norm_out; ConvW conv_out; ConvW skip;
    bool has_skip = false, has_embed = false;
    int cin = 0, cout = 0, temb_index = -1;
};
struct MhaW { LinW q, k, v, out; };
struct SpatialW {  // SpatialTransformer + TransformerBlock (unet/mod.rs:454-527)
    NormW norm; ConvW proj_in, proj_out; NormW ln1, ln2, ln3; MhaW attn1, attn2; LinW geglu_proj, mlp_lin;
    int c = 0, ctx_index = -1;
};
enum BlockKind { BK_CONV, BK_DOWN, BK_RES, BK_RES_ST, BK_RES_UP, BK_RES_ST_UP };
struct UBlock { BlockKind kind; int cin, cout; ConvW conv; ResW res; SpatialW st; ConvW up; };
struct VaeAttnW { NormW norm; ConvW q, k, v, proj_out; int c = 0; };
struct DecBlockW { ResW res[3]; ConvW upsampler; bool has_up = false; int cin = 0, cout = 0; };

// CLIP text encoder block (ResidualDecoderAttentionBlock, clip/mod.rs:95-114); q/k/v weights and biases are
// packed for one N = 3C GEMM
struct ClipBlockW { NormW attn_ln, mlp_ln; LinW q, k, v, out, fc1, fc2; };

struct WeightEntry {
    std::string name;
    int kind;  // 0 conv (OIHW), 1 linear ([in,out]), 2 vector, 3 alphas (host)
    int ndim;
    int64_t dims[4];
    float** dst;  // where the device pointer lives (null for alphas)
    int wdt = 0;  // storage type of the packed weight: 0 fp32, 1 bf16
    int group = 0;  // 0: hot path (required); 1: CLIP text encoder, 2: VAE encoder (each optional as a whole)
    float** dst8 = nullptr;   // precision = 2: where the MXFP8 copy of a conv weight and its scales go (null: none)
    float** dsts = nullptr;
    float pre_scale = 1.f;    // the tensor is multiplied by this in fp32 before it is packed (bf16 / MXFP8 query projections: attn_bf16_q_scale)
    bool set = false;
};

// Per-module scalar / 2-vector files of the dump tree that are not tensors (python/save.py:23-68): `store` != null: the value
// is honoured (a norm's eps); otherwise it must equal `expect` (the hyper-parameters this engine hard-wires).
struct MetaEntry { std::string name; int n; float expect[2]; float* store; };

struct TileChoice { int cfg; int splits; };

class Engine {
public:
    explicit Engine(const sdmi_config& cfg);
    ~Engine();
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // weights
    void set_weight(const char* name, const float* data, int ndim, const int64_t* dims);
    void load_weights_dir(const char* dir);
    void load_weights_mpk(const char* path);
    void load_weights_packed(const float* data, size_t n_floats, int groups);
    size_t packed_size(int groups) const;
    void finalize_weights();
    const std::vector<WeightEntry>& entries() const { return entries_; }

    // hot path (device pointers, reference layouts)
    void unet_forward_dev(const float* x_nchw, int t, const float* context, int n, int T, float* out_nchw);
    // CLIP::forward (clip/mod.rs:56-75): int32 tokens [n, T] on the device -> [n, T, ctx_dim] fp32 (both precisions)
    void clip_forward_dev(const int32_t* tokens, int n, int T, float* out);
    bool clip_ready() const { return clip_ready_; }
    // Autoencoder::encode_image (autoencoder/mod.rs:60-66): img [n,3,8h,8w] NCHW -> latent mean [n,4,h,w] NCHW (device pointers)
    void encode_image_dev(const float* img_nchw, int n, float* latent_nchw);
    void sample_latent_dev(const float* context, int n, int T, const float* uncond, int Tu, double scale,
                           size_t n_steps, const float* init_latent, float* latent_out);
    void decode_latent_dev(const float* latent_nchw, int n, float in_scale, float* img_nchw, uint8_t* rgb_u8);
    void qkv_attention_dev(const float* q, const float* k, const float* v, const float* mask, int mask_ld, int n,
                           int nq, int nk, int n_state, int n_head, float* out);

    // operator-level (device pointers, reference layouts)
    void op_group_norm(const float* x, const float* gamma, const float* beta, int n, int c, int h, int w, int groups,
                       float eps, bool silu, float* out);
    void op_group_norm_fp8(const float* x, const float* gamma, const float* beta, int n, int c, int h, int w, int groups, float eps,
                           bool silu, float* out);
    void op_layer_norm(const float* x, const float* gamma, const float* beta, int rows, int c, float eps, float* out);
    void op_conv2d(const float* x, const float* w, const float* bias, int n, int cin, int h, int wd, int cout, int k,
                   int stride, int pad, int ups, float* out);
    void op_linear(const float* x, const float* w, const float* bias, int rows, int cin, int cout, float* out);
    void op_geglu(const float* proj, int rows, int hidden, float* out);
    void op_geglu_forward(const float* x, const float* wt, const float* bias, int rows, int cin, int hidden, float* out);
    void op_timestep_embedding(int t, int dim, float* out);
    double bench_conv(int n, int cin, int h, int w, int cout, int k, int stride, int ups, int tile_cfg, int splitk,
                      int iters);

    double bench_attention(int n, int nq, int nk, int n_state, int n_head, int iters);
    void set_option(const std::string& key, const std::string& value);
    void sync();
    // Every C-ABI entry point runs inside a Call: the constructor orders the engine's stream behind the caller's work
    // (dev_inputs: the arguments are device buffers produced on another stream), finish() waits for the results and
    // records the call statistics, and a Call destroyed by an exception returns every pool block the call allocated.
    void begin_call(bool dev_inputs = false);
    void end_call();
    void abort_call() noexcept;
    struct Call {
        Engine& e; bool done = false;
        Call(Engine& e_, bool dev_inputs = false) : e(e_) { e.begin_call(dev_inputs); }
        void finish() { e.end_call(); done = true; }
        ~Call() { if (!done) e.abort_call(); }
        Call(const Call&) = delete;
        Call& operator=(const Call&) = delete;
    };
    // the stream the caller's device buffers are produced / consumed on (sdmi_set_stream); has_user_stream_ false:
    // *_dev entry points synchronise the whole device on entry instead
    void set_user_stream(hipStream_t s, bool enable) { user_stream_ = s; has_user_stream_ = enable; }
    double last_ms = 0;
    long long last_kernels = 0;
    double last_flops = 0;

    hipStream_t stream() const { return stream_; }
    DevPool& pool() { return pool_; }
    const sdmi_config& config() const { return cfg_; }
    int latent_h() const { return cfg_.latent_h; }
    int latent_w() const { return cfg_.latent_w; }

    // RAII helper for pool scratch
    struct Buf {
        Engine* e; void* p;
        Buf(Engine* e_, size_t bytes) : e(e_), p(e_->pool_.alloc(bytes)) {}
        ~Buf() { if (p) e->pool_.free(p); }
        Buf(const Buf&) = delete;
        Buf& operator=(const Buf&) = delete;
        float* f() const { return reinterpret_cast<float*>(p); }
    };

private:
    void destroy() noexcept;
    // batched weight staging (engine.cpp "weights")
    struct Stager;
    std::unique_ptr<Stager> stager_;
    bool arena_done_[3] = {false, false, false};
    // precision = 0: every packed fp32 weight also exists as three bf16 planes (k_gemm3x.hip) in a parallel arena; the planes of
    // the weight at byte offset o of arena g start at offset 3 o / 2 of split arena g (6 bytes per weight instead of 4), so
    // weights that are adjacent rows of one GEMM (q | k | v) stay adjacent
    char* arena_base_[3] = {nullptr, nullptr, nullptr};
    size_t arena_bytes_[3] = {0, 0, 0};
    char* split_base_[3] = {nullptr, nullptr, nullptr};
    struct SplitRegion { char* base; size_t bytes; char* planes; };
    std::vector<SplitRegion> split_regions_;
    const void* split_planes(const float* bt) const;
    // weight planes of a tensor with `rows` rows: 16-row fragment groups (kernels.hpp, ConvGemm::b3_grouped) whenever the rows fill whole groups.  Packer and launches
    // apply the same rule; sub-views of a packed tensor (q | k | v) start and end on multiples of 16 rows.
    bool b3_grouped(long long rows) const { return opt_b3_grouped_ != 0 && rows % 16 == 0; }
    // operator-level calls (op_conv2d, op_linear, bench_conv ...) pack their weight into a pool buffer: this gives it planes for
    // the duration of the call, so that those calls run the kernels the model runs
    struct TempSplit {
        Engine* e; const float* bt; void* planes = nullptr;
        TempSplit(Engine* e_, const float* bt_, long long rows, long long K);
        ~TempSplit();
        TempSplit(const TempSplit&) = delete;
        TempSplit& operator=(const TempSplit&) = delete;
    };
    const float* temp_split_bt_ = nullptr;
    const void* temp_split_planes_ = nullptr;
    char* stage_reserve(size_t bytes, size_t* offset, int* half);
    void stage_commit(WeightEntry& e, size_t offset, int half);
    void upload_weight(WeightEntry& e, const float* data);
    void stager_release();
    void ensure_arena(int group);
    bool set_meta(const std::string& name, const float* values, size_t n);
    void add_meta(const std::string& name, int n, float e0, float e1, float* store);
    std::vector<MetaEntry> meta_;
    std::map<std::string, int> meta_index_;
    // model definition
    void add_entry(const std::string& name, int kind, std::initializer_list<int64_t> dims, float** dst, int wdt = 0);
    int cur_group_ = 0;  // weight group add_entry assigns (build_model switches it to 1 for the CLIP section)
    void build_model();

    // primitive ops on device activations (NHWC)
    Act new_act(int n, int h, int w, int c, int dt = -1);  // dt -1: the engine's activation type
    // fp32 engines: what = 1 fp32 only, 2 planes only, 3 both (c % 32 == 0 for planes)
    Act new_act3(int n, int h, int w, int c, int what);
    void release(Act& a);
    // pad_br: zero padding on the bottom / right only (PaddingCfg::new(0, 1, 0, 1), the VAE encoder's downsampler)
    void probe_report(void* pb_dev, size_t max_blocks, int n, int cin, int h, int w, int cout, int k, int tile_cfg, int splitk);
    void conv(const ConvW& w, const Act& x, Act& y, int stride, int ups, const float* rowvec, int rowvec_stride,
              const Act* resid, bool pad_br = false);
    static Act slice(const Act& parent, int c_off, int c);   // channel-slice view
    // A3 / C3: the input / output as three bf16 planes (dense rows: 192 bytes per 32 channels); A and / or C may then be null
    void gemm(const float* A, int a_rows, const float* bt, const float* bias, int cin, int cout, float* C, int ldc,
              const float* resid, int ldr, int dt = -1, int out_mode = 0, const void* A3 = nullptr, void* C3 = nullptr);
    // fp32 engine, option gemm_planes: does the GEMM cin -> cout take its activations as planes (k_gemm3p.hip)?
    bool plane_gemm(int cin, int cout) const { return !bf16_ && opt_gemm_planes_ != 0 && opt_gemm_f32s_ != 0 && cin % 32 == 0 && cout >= 32; }
    void launch_gemm(ConvGemm& p, int in_dt, int force_cfg = -1, int force_splits = 0);
    int edt() const { return bf16_ ? 1 : 0; }
    size_t esz() const { return bf16_ ? 2 : 4; }
    // element-wise pointer advance on an activation of type dt
    static float* adv(const float* p, long long elems, int dt) { return (float*)((char*)const_cast<float*>(p) + elems * (dt ? 2 : 4)); }
    TileChoice choose_tile(int M, int N, int kt_total, bool allow_x = false, bool allow_s = false) const;   // cfg >= 100: k_gemm2x.hip tile cfg - 100, >= 200: k_gemm3x.hip tile cfg - 200
    void group_norm(const NormW& w, const Act& x, Act& y, bool silu);
    // precision = 2: GroupNorm(+SiLU) writing MXFP8, and the 3x3 convolution that consumes it (k_fp8.hip)
    ActQ new_actq(int n, int h, int w, int c);
    void release(ActQ& a);
    void group_norm_fp8(const NormW& w, const Act& x, ActQ& y, bool silu);
    // stride / ups as conv(); 3x3 (pad 1) or 1x1 (pad 0) -- whatever was packed as MXFP8 (ConvW::bt8)
    void conv_fp8(const ConvW& w, const ActQ& x, Act& y, const float* rowvec, const Act* resid, int stride = 1, int ups = 0);
    bool use_fp8(const ConvW& w, const Act& x) const;
    // option fp8_linear: the layers beyond the ResBlock 3x3 convolutions -- Linear layers, 1x1 / up / down convolutions
    bool use_fp8_wide(const float* bt8, long long rows) const { return fp8_ && opt_fp8_convs_ && opt_fp8_linear_ && bt8 && rows >= opt_fp8_min_rows_; }
    void launch_fp8(ConvGemm& p, double flops);                                   // tile / split-K choice + launch (+ reduce) of conv_gemm_fp8x_kernel
    void gemm_fp8(const ActQ& x, const LinW& w, int n_rows_w, void* C, int ldc, const float* resid, int ldr);   // C[rows][n_rows_w] = x W^T + b (+ resid), bf16 out
    void conv_raw(const ConvW& w, const Act& x, Act& y, int stride, int ups);     // conv() or, at precision = 2, quantize() + conv_fp8()
    void quantize(const Act& x, ActQ& y);                                         // bf16 activation -> MXFP8 (quantize_bf16_fp8_kernel)
    void layer_norm_fp8(const NormW& w, const float* x, long long rows, ActQ& y);
    ActQ new_rowsq(long long rows, int c) { return new_actq(1, 1, (int)rows, c); }
    void layer_norm(const NormW& w, const float* x, long long rows, float* y, int dt = -1, void* y3 = nullptr);   // y3: output as planes instead of y
    // GEGLU::forward (unet/mod.rs:579-591): out[rows, hidden] = (x W + b)[:, :hidden] * gelu((x W + b)[:, hidden:]); bt is the
    // packed [2 hidden][cin] weight.  Fused into a large-tile GEMM when possible, else GEMM into `proj_scratch` + gate kernel.
    void gemm_geglu(const float* x, long long rows, const float* bt, const float* bias, int cin, int hidden, float* out, int dt,
                    const void* x3 = nullptr, void* out3 = nullptr);
    void attention(const float* q, int ldq, long long q_bs, const float* k, int ldk, long long k_bs, const float* v,
                   int ldv, long long v_bs, float* o, int ldo, long long o_bs, int n, int nq, int nk, int n_head,
                   int d_head, const int* kv_len_dev, const int* kv_len_host, const float* mask, int mask_ld, int dt = -1, void* o3 = nullptr, bool q_log2 = false);
    // ONE rule for "the query arrives multiplied by attn_bf16_q_scale(d_head)": bf16 storage at the head dims the fused bf16 kernel serves.  The weight loader
    // folds the factor into the query projection by it, the fp32 -> bf16 conversions of the operator entry points apply it by it, and every attention() call
    // passes it as q_log2 -- the kernel launcher refuses a bf16 call that does not state it.
    static bool q_prescaled(int dt, int d_head) { return dt != 0 && attn_bf16_q_is_log2(d_head); }

    // composite blocks
    void res_block(const ResW& w, const Act& x, Act& y, int step);
    void spatial_transformer(const SpatialW& w, const Act& x, Act& y);
    void vae_attn(const VaeAttnW& w, const Act& x, Act& y);

    // UNet driver
    void unet_prepare(const float* ctx_packed, int nb, int t_max, const int* kv_len_host, const std::vector<int>& ts);
    void unet_release();
    void unet_run(const float* x_nhwc, int nb, int step, float* out_nhwc, bool cfg_pair = false);
    void decode_one(const float* z_nhwc, int n, Act& img);

    void count_kernel(double flops = 0) { ++n_kernels_; flops_ += flops; }
    // roctx ranges (option "roctx=1"; rocprofv3 --marker-trace shows them): one per DDIM step, UNet block, ResBlock,
    // SpatialTransformer and VAE stage.  libroctx64 is opened lazily, like RCCL.
    struct Range {
        Engine* e;
        Range(Engine* e_, const char* name) : e(e_->roctx_on_ ? e_ : nullptr) { if (e) e->roctx_push(name); }
        Range(Engine* e_, const std::string& name) : Range(e_, name.c_str()) {}
        ~Range() { if (e) e->roctx_pop(); }
        Range(const Range&) = delete;
        Range& operator=(const Range&) = delete;
    };
    bool roctx_on_ = false;
    void roctx_enable(bool on);
    void roctx_push(const char* name);
    void roctx_pop();
    void check_batch(int n) const {
        if (cfg_.max_batch > 0 && n > cfg_.max_batch)
            throw Error(SDMI_ERR_INVALID, "batch of " + std::to_string(n) + " exceeds sdmi_config.max_batch = " + std::to_string(cfg_.max_batch));
    }

public:
    // per-kernel-class timing (option "profile=1"): HIP events around every launch on the
    // engine's stream, accumulated per class.  Used by bench.py for the roofline line.
    enum ProfClass { PC_CONV_GEMM, PC_SPLITK_REDUCE, PC_ATTENTION, PC_GROUP_NORM, PC_LAYER_NORM, PC_CONV_FP8, PC_CONV_SPLIT, PC_SPLIT_ROWS, PC_OTHER, PC_GEGLU, PC_COUNT };
    struct ProfStat { double ms = 0; long long launches = 0; double flops = 0; double bytes = 0; };
    ProfStat prof_[PC_COUNT];
    void prof_flush();
    void prof_calibrate();
    double prof_overhead_ms_ = 0;    // what an empty event pair reads (subtracted from every sample)
    void prof_reset() { prof_flush(); for (auto& p : prof_) p = ProfStat{}; prof_tags_.clear(); }
    // option "profile=2": the same samples also accumulated per launch TAG (class + shape + tile choice), option dump_profile_tags writes them:
    // where inside a class the time goes (tools/shape_times.py)
    std::map<std::string, ProfStat> prof_tags_;

private:
    struct ProfScope {
        Engine* e; int cls; double flops, bytes; int n_launch; hipEvent_t a = nullptr, b = nullptr; int tag = -1;
        ProfScope(Engine* e_, int cls_, double flops_ = 0, double bytes_ = 0, int n_launch_ = 1);    // n_launch: kernels inside the scope (GroupNorm: statistics + apply)
        ~ProfScope();
        void set_tag(const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // no-op unless profile=2
    };
    struct ProfPending { int cls; hipEvent_t a, b; double flops, bytes; int n_launch; int tag; };
    bool prof_tagging_ = false;
    std::vector<std::string> prof_tag_names_;
    std::map<std::string, int> prof_tag_ids_;
    hipEvent_t prof_event();
    bool profiling_ = false;
    std::vector<hipEvent_t> prof_free_;
    std::vector<ProfPending> prof_pending_;

    sdmi_config cfg_;
    bool bf16_ = false;  // precision >= 1: bf16 activations / weights, fp32 accumulate
    bool fp8_ = false;   // precision = 2: additionally the ResBlock / ResnetBlock 3x3 convs in MXFP8 (k_fp8.hip)
    int opt_fp8_convs_ = 1;          // 0: run the fp8-capable convs on the bf16 kernels (A/B, accuracy comparison)
    int opt_fp8_min_rows_ = 1024;    // GEMMs with fewer output rows stay bf16 (256-row tiles need rows to fill the chip)
    int opt_fp8_tile_ = -1;
    int opt_gn32_min_wgs_ = 256 | ((64 + 1) << 16);   // precision = 0, GroupNorm launch geometry (k_norm.hip gn_geom): low 16 bits = at least this many workgroups in the APPLY pass over
                                     // the call's samples (a batch-1 tensor cut by size alone leaves CUs without a workgroup); bits 16.. = 1 + the same for the STATISTICS pass, which
                                     // is cut coarser (every apply workgroup merges all chunk partials).  Options gn32_min_wgs / gn32_stats_min_wgs / gn32_stats_chunk_kb;
                                     // measured at batch 1: GroupNorm class 21.3 -> 17.3 ms per image (profiles/r05d, r05f, r05o, r05p)
    GnTune gn_tune_;                 // launch geometry of the bf16 / MXFP8 GroupNorm passes (kernels.hpp; options gn_target_wgs, gn_max_threads, gn_unroll)
    int opt_b3_grouped_ = 1;         // precision 0: weight planes in 16-row fragment groups (1 KiB DMA pieces, sequential per group); 0 = row-major planes.  Before the weights are loaded.
    int opt_attn_pack_tail_ = 3;     // fp32 attention (k_attn_split.hip): bit 0: d = 40's columns 32..39 as a packed k step / packed output tile; bit 1: scores in log2 units with the
                                     // reference maximum as accumulator input and the row sum from a ones column (A/B, tests)
    int opt_attn_kv_prefer8_ = 1;    // ... and, for k_attn_split.hip, as many slices as let its 8-wave form fill the chip (A/B switch)
    int opt_attn_kv_splits_ = 0;     // fp32 attention: key slices + merge launch where the query-tile grid leaves CUs idle (Engine::attention): 0 = automatic, 1 = never, S = forced
    int opt_cfg_share_ = 1;          // sample_latent: the part of the UNet in front of the first cross attention is computed once for the two identical halves of a CFG step (unet_run)
    int opt_op_resid_ = 0;           // tests: op_conv2d / op_linear add their input as the residual (cin == cout) through the GEMM epilogue
    int opt_fp8_ops_ = 0;            // tests: op_linear / op_layer_norm / op_geglu run the fp8_linear path's kernels (outputs dequantised)
    int opt_fp8_linear_ = 0;         // precision = 2: 0 (default: the accuracy budget of 6e-2 final-latent relative RMS) = MXFP8 on the ResBlock / ResnetBlock 3x3
                                     // convolutions only; 1 = also the transformer blocks' Linear layers and the 1x1 / up / down convolutions (8.1e-2)
    hipStream_t stream_ = nullptr;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr, ev_user_ = nullptr;
    hipStream_t user_stream_ = nullptr;
    bool has_user_stream_ = false;
    bool call_dev_ = false;
    unsigned long long call_mark_ = 0;
    DevPool pool_;
    std::vector<WeightEntry> entries_;
    std::map<std::string, int> entry_index_;
    std::vector<void*> weight_allocs_;
    bool finalized_ = false;
    std::vector<float> alphas_;

    // UNet
    LinW lin1_time_, lin2_time_;
    std::vector<UBlock> in_blocks_, out_blocks_;
    ResW mid_res1_, mid_res2_;
    SpatialW mid_st_;
    NormW unet_norm_out_;
    ConvW unet_conv_out_;
    std::vector<ResW*> res_list_;       // index = temb_index
    std::vector<SpatialW*> st_list_;    // index = ctx_index
    // VAE decoder
    ConvW post_quant_, dec_conv_in_, dec_conv_out_;
    ResW dec_mid1_, dec_mid2_;
    VaeAttnW dec_attn_;
    DecBlockW dec_blocks_[4];
    NormW dec_norm_out_;
    // CLIP text encoder (optional weight group)
    float* clip_tok_ = nullptr;
    float* clip_pos_ = nullptr;
    std::vector<ClipBlockW> clip_blocks_;
    NormW clip_ln_;
    bool clip_ready_ = false;
    // VAE encoder (optional weight group)
    struct EncBlockW { ResW res[2]; ConvW down; bool has_down = false; int cin = 0, cout = 0; };
    ConvW enc_conv_in_, enc_conv_out_, quant_conv_;
    EncBlockW enc_blocks_[4];
    ResW enc_mid1_, enc_mid2_;
    VaeAttnW enc_attn_;
    NormW enc_norm_out_;
    bool enc_ready_ = false;

    // per-call UNet state
    struct UNetState {
        int nb = 0, t_max = 0, steps = 0;
        std::vector<float*> temb;   // per ResBlock [steps][cout]
        std::vector<float*> kc, vc; // per SpatialTransformer [nb][t_max][c]
        int* kv_len_dev = nullptr;
        std::vector<int> kv_len_host;
        std::vector<void*> owned;
    } us_;

    // options
    int opt_force_tile_ = -1;
    int opt_force_splits_ = 0;
    int opt_resid_acc_ = 3;     // precision >= 1, large-tile kernels without split-K (ConvGemm::resid_acc): bit 0 = the residual, bit 1 = bias + time-embedding row are the accumulators'
                                // initial value, loaded in front of the k loop; 0 = added by the epilogue (round 5)
    int opt_attn_bf16_ = 1;
    static constexpr int kAttnBf16VariantDefault = 7;
    int opt_attn_bf16_variant_ = kAttnBf16VariantDefault;   // k_attn_bf16.hip (AttnParams::variant): bit 0 = 4-wave workgroups, two per CU; bit 1 / 2 = 64 query rows per wave (d = 40) on 8- / 4-wave workgroups; 0x100 = whatever the grid (tests)
    int opt_geglu_fuse_ = 1;    // GEGLU gate in the projection GEMM's epilogue: 0 never, 1 where there are >= 4 rounds of tiles, 2 / 3 always (256x128 / 256x256 tiles; tests)
    int opt_attn_split_ = 1;    // precision = 0: 1 = d_head 40 / 80 attention on the bf16 matrix pipe with three-way split operands (k_attn_split.hip)
    int opt_gemm_f32s_ = 1;     // precision = 0: 1 = fp32 GEMMs on the bf16 matrix pipe (three-way operand split, k_gemm3x.hip) where faster
    static constexpr int kGemm3xVariantDefault = 2;
    int opt_gemm3x_variant_ = kGemm3xVariantDefault;     // k_gemm3x.hip: bit 0: DMA in one block per k tile; bit 1: scalar residual subtractions (+0.7 %); bit 2: two LDS stages on the 128-row tiles
                                     // (default three: +5..10 % on long K); bit 4: s_setprio 1 for waves 4-7 (measured: no gain)
    static constexpr int kGemmPlanesDefault = 1;
    int opt_gemm_planes_ = kGemmPlanesDefault;   // precision = 0: k_gemm3p.hip (activations as bf16 planes too, no split in the k loop): 0 never, 1 every launch that would take a k_gemm3x.hip tile,
                                // 2 = A/B switch (tests): every launch that chose a k_gemm3x.hip tile runs on the nearest k_gemm3p.hip tile, its fp32 input converted by split3_rows_kernel in front of it
    int opt_conv3_reuse_ = 1;   // precision >= 1: 1 = 3x3 / stride-1 convolutions that chose the 256 x 320 / 256 x 256 tile run on k_gemm_bf16t.hip (one staged activation tile per kernel row)
    int opt_gemm_probe_ = 0;    // bench_conv: 1 = one extra launch with per-workgroup phase stamps (ConvGemm::probe), summary on stderr
    unsigned long long* probe_buf_ = nullptr;
    int opt_bench_cold_ = 0;    // bench_conv: 1 = evict the weights from the Infinity Cache between timed launches (what a layer sees inside the model)
    int opt_gemm_x32_ = 1;      // precision = 0: 1 = large-tile LDS-DMA fp32 GEMM (k_gemm2x.hip) where measured / modelled faster
    static constexpr int kGemmBf16xVariantDefault = 5;
    int opt_gemm_bf16x_variant_ = kGemmBf16xVariantDefault; // precision >= 1, k_gemm_bf16x.hip / k_gemm_bf16t.hip: bit 2 (round 6) = waves 4 - 7 issue their DMA pieces between a tile's two k steps (one-tile forms and the kernel-row convolution); bit 0 = persistent tile loop (launches without split-K or residual and with more tiles than CUs;
                                     // bit-identical results, +0.4 ... 0.7 % per image: profiles/r05a_*)
    int opt_gemm_bf16x_ = 1;    // precision = 1: 1 = large-tile LDS-DMA GEMM where the cost model prefers it; 0 = never
    void* zero_page_ = nullptr;
    TileChoice choose_tile_p(int M, int N, int kt_total, bool even_ni_only) const;   // k_gemm3p.hip tiles (300 + x)
    TileChoice choose_tile_bf16(int M, int N, int kt_total) const;   // cfg >= 100: k_gemm_bf16x.hip tile cfg - 100     // precision = 1: 1 = bf16 matrix-core attention, 0 = bf16 storage widened onto the fp32 kernel  // 1: attn2_kernel, 0: attn_f32_kernel
    std::map<std::string, TileChoice> tuned_;        // fp32 kernels: "M,N,K" -> (tile cfg, split-K)
    std::map<std::string, TileChoice> tuned_bf16_;   // bf16 kernels; cfg >= 100 = k_gemm_bf16x.hip tile
    std::map<std::string, TileChoice> tuned_p_;      // fp32 shapes whose activations arrive as planes: k_gemm3p.hip tiles (300 + x) only
    std::map<std::string, TileChoice> tuned_mfma_;   // fp32 shapes measured with the fp32-MFMA kernels only (gemm_f32s=0)
    bool record_shapes_ = false;
    std::map<std::string, long long> shape_counts_;  // "n,cin,h,w,cout,k,stride,ups" -> launches
    std::map<std::string, long long> choice_counts_;   // "M,N,K cfg=.. splits=.." -> launches (record_shapes)

    long long n_kernels_ = 0;
    double flops_ = 0;
};

void set_last_error(const std::string& msg);   // the thread-local message behind sdmi_last_error() (sdmi_capi.cpp)

}  // namespace sdmi

// the opaque handle of include/sdmi.h
struct sdmi_ctx {
    sdmi::Engine* engine;
};
