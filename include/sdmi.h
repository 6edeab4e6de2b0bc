/*
 * sdmi.h -- C ABI of libsdmi.so: the MI355X (gfx950) Stable Diffusion v1.4
 * sampling hot path (UNet DDIM+CFG loop and VAE decoder) that sits behind the
 * `StableDiffusion::sample_image` surface of Gadersd/stable-diffusion-burn.
 *
 * Every entry point below names the reference interface it replaces
 * (file:line relative to the reference repo).  The reference has no FFI of its
 * own -- its "plugin" seam is the Burn `Backend` type parameter
 * (src/bin/sample/main.rs:59-83) plus the commented-out operator-override
 * trait in src/backend.rs:4-84 -- so this header is what a Rust shim
 * (ffi/sdmi.rs) binds with `extern "C"`.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types
 *  - host-pointer functions take row-major fp32 in the REFERENCE's logical
 *    layouts (NCHW images/latents, [n, tokens, channels] sequences); the
 *    library copies in/out and blocks until results are in host memory.
 *    NHWC and packed weights are internal.
 *  - *_dev functions take DEVICE pointers (same logical layouts).  The context
 *    works on a private non-blocking HIP stream, so on entry it must be ordered
 *    behind whatever produced those buffers: if sdmi_set_stream() named the
 *    caller's stream, the context waits for an event on it (and makes that
 *    stream wait for the results on return); otherwise the call starts with a
 *    hipDeviceSynchronize().  All entry points return after the results are
 *    complete (they block on the context's stream).
 *  - every function returns 0 (SDMI_OK) or a negative sdmi_status; the
 *    message is available from sdmi_last_error() (thread-local).  The
 *    reference's hot path is infallible by type and panics on shape errors
 *    (stablediffusion/mod.rs:86, unet/mod.rs:134); the Rust shim panics on a
 *    non-zero status to keep that contract.
 *  - a context is not re-entrant: one call at a time per context.
 *  - the caller owns every in/out buffer; the context owns device weights,
 *    activation pool and stream.  Nothing returned needs freeing but the ctx.
 */
#ifndef SDMI_H
#define SDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdmi_ctx sdmi_ctx;

typedef enum sdmi_status {
    SDMI_OK = 0,
    SDMI_ERR_INVALID = -1,     /* bad argument / shape (reference: panic)      */
    SDMI_ERR_HIP = -2,         /* a hip* call failed                           */
    SDMI_ERR_WEIGHTS = -3,     /* missing / mis-shaped weight tensor           */
    SDMI_ERR_IO = -4,          /* weight file / directory unreadable           */
    SDMI_ERR_UNSUPPORTED = -5, /* valid in the reference, not built here yet   */
    SDMI_ERR_STATE = -6        /* call order (e.g. forward before finalize)    */
} sdmi_status;

/* Model / device configuration.  Defaults (sdmi_default_config) are the
 * reference's hard-coded hyper-parameters: UNetConfig::init unet/mod.rs:36-92,
 * AutoencoderConfig::init autoencoder/mod.rs:30-36, latent 4x64x64
 * stablediffusion/mod.rs:116.  Tests shrink them; nothing else should. */
typedef struct sdmi_config {
    int32_t device;          /* HIP device ordinal                              */
    int32_t model_channels;  /* 320                                             */
    int32_t n_head;          /* 8                                               */
    int32_t ctx_dim;         /* 768  (CLIP text width)                          */
    int32_t latent_h;        /* 64   (image = 8x)                               */
    int32_t latent_w;        /* 64                                              */
    int32_t vae_ch;          /* 128  (decoder channels 4c,4c,2c,c)              */
    int32_t max_batch;       /* largest n a call may pass; 0 = no limit          */
    int32_t precision;       /* 0 = fp32; 1 = bf16 storage, fp32 accumulate; 2 = 1 + MXFP8 operands for the ResBlock / ResnetBlock 3x3 convolutions (20-step latent 5.2e-2 relative RMS
                              * of the exact one; option "fp8_linear=1": also the transformer blocks' Linear layers and the 1x1 / up / down convolutions, 8.1e-2 for +5 %).
                              * ATTENTION IS NOT ON FP8 OPERANDS at any precision (BASELINE.json configs[4] names "fp8 conv+attn"): qkv_attention runs bf16 at precision 1 and 2 --
                              * formally dropped, DESIGN.md section 8: Q K^T contracts over d_head = 40 (no gain on the K = 128 MX instruction), P V would return at most 2.5 % of an
                              * image (measured ablation), at the price of an e4m3 P with block scales inside the softmax loop. */
    /* CLIP text encoder, CLIPConfig::new(49408, 768, 12, 77, 12) stablediffusion/mod.rs:29;
     * its width is ctx_dim.  clip_layers = 0 builds a context without it.              */
    int32_t clip_layers;     /* 12                                              */
    int32_t clip_heads;      /* 12                                              */
    int32_t clip_vocab;      /* 49408                                           */
    int32_t clip_ctx;        /* 77                                              */
    int32_t reserved[3];
} sdmi_config;

int sdmi_default_config(sdmi_config* cfg);

/* ---- lifecycle ------------------------------------------------------------ */
int sdmi_create(sdmi_ctx** out, const sdmi_config* cfg);
void sdmi_destroy(sdmi_ctx* ctx);
const char* sdmi_last_error(void);
int sdmi_synchronize(sdmi_ctx* ctx);
/* Names the HIP stream (a hipStream_t passed as void*; NULL = the legacy default stream) on which the caller
 * produces the inputs and consumes the outputs of the *_dev entry points; enable = 0 returns to the default
 * (device-wide synchronisation on entry).  The reference has no counterpart: Burn tensors are ordered by the
 * backend's own queue. */
int sdmi_set_stream(sdmi_ctx* ctx, void* hip_stream, int32_t enable);
/* library / build identification, e.g. "sdmi 0.1 gfx950 fp32" */
const char* sdmi_version(void);

/* ---- weights -------------------------------------------------------------
 * Names are the reference's npy-dump tree paths (src/model/unet/load.rs:217-305,
 * src/model/autoencoder/load.rs:135-157, src/model/stablediffusion/load.rs:20-24),
 * e.g. "unet/input_blocks/rt1/res/conv_in/weight", "autoencoder/post_quant_conv/bias",
 * "alphas_cumprod".  Shapes are the reference's: Conv2d weight [Cout,Cin,kh,kw],
 * Linear weight [in,out] (python/save.py:19), norms [C].  Replaces
 * load_stable_diffusion (stablediffusion/load.rs:16-33) for the hot-path
 * subset (UNet, VAE decoder + post_quant_conv, alphas_cumprod). */
int sdmi_set_weight(sdmi_ctx* ctx, const char* name, const float* data, int32_t ndim, const int64_t* dims);
/* number of tensors the configured model needs / name + shape of the i-th */
int sdmi_weight_count(sdmi_ctx* ctx);
int sdmi_weight_info(sdmi_ctx* ctx, int32_t index, const char** name, int32_t* ndim, int64_t dims[4]);
/* reads the npy-dump directory written by the reference's python/ exporters
 * (format: src/model/load.rs:17-28 -- 1-D float32 .npy whose first D values
 * are the shape). */
int sdmi_load_weights_dir(sdmi_ctx* ctx, const char* dump_dir);
/* load_stable_diffusion_model_file (src/bin/sample/main.rs:27-34): reads a Burn NamedMpkFileRecorder<FullPrecisionSettings>
 * record (the reference's "SDv1-4.mpk") natively -- a MessagePack walker over the memory-mapped file, tensors staged
 * straight from the mapping.  The burn 0.14 layout it assumes is spelled out in csrc/mpk_reader.hpp (UNPINNED against a
 * real record: none exists offline).  Tensors the configured model lacks are skipped; call sdmi_finalize_weights after. */
int sdmi_load_weights_mpk(sdmi_ctx* ctx, const char* mpk_path);
/* Host only (no context): the tensors of a record as text, one "name<TAB>d0,d1,..<TAB>file offset" line each (dump-tree
 * names), after a "# format=.. float=.." line.  *needed = bytes incl. the terminator; out may be NULL to query it. */
int sdmi_mpk_list(const char* mpk_path, char* out, size_t capacity, size_t* needed);
/* One flat image of every tensor: `data` holds, for each entry i of the configured model in
 * sdmi_weight_info() order and restricted to the weight groups selected by `groups` (bit 0: hot path = UNet,
 * VAE decoder, alphas_cumprod; bit 1: CLIP; bit 2: VAE encoder), the tensor's fp32 values in the reference's
 * layout, back to back, no headers.  `n_floats` must equal the sum of their sizes (sdmi_packed_size).  Staged
 * through one pinned buffer and one stream: the batched replacement of load_stable_diffusion's ~1100 file reads
 * (stablediffusion/load.rs:16-33, model/load.rs:17-160). */
int sdmi_load_weights_packed(sdmi_ctx* ctx, const float* data, size_t n_floats, int32_t groups);
/* number of floats sdmi_load_weights_packed expects for `groups` */
int64_t sdmi_packed_size(sdmi_ctx* ctx, int32_t groups);
/* packs everything into the device layouts; fails listing the first missing tensor */
int sdmi_finalize_weights(sdmi_ctx* ctx);

/* ---- hot path, host pointers ------------------------------------------------ */

/* UNet::forward (src/model/unet/mod.rs:109-143).
 * x [n,4,h,w] NCHW, t scalar timestep (the reference passes a 1-element Int
 * tensor shared by the batch), context [n,T,ctx_dim] -> out [n,4,h,w]. */
int sdmi_unet_forward(sdmi_ctx* ctx, const float* x, int32_t t, const float* context,
                      int32_t n, int32_t T, float* out);

/* StableDiffusion::sample_latent (src/model/stablediffusion/mod.rs:102-160),
 * DDIM eta=0 with classifier-free guidance (forward_diffuser :162-192).
 * context [n,T,ctx_dim]; uncond [Tu,ctx_dim] (broadcast over the batch);
 * init_latent [n,4,h,w] = x_T (the reference draws it from an unseeded backend
 * RNG, :115-121; here it is an explicit input) or NULL to draw N(0,1) from
 * `seed` (image i uses stream seed+i); latent_out [n,4,h,w]. */
int sdmi_sample_latent(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T,
                       const float* uncond, int32_t Tu, double scale, size_t n_steps,
                       const float* init_latent, uint64_t seed, float* latent_out);

/* Autoencoder::decode_latent (src/model/autoencoder/mod.rs:68-71):
 * latent [n,4,h,w] (already divided by 0.18215 by the caller, as in
 * latent_to_image) -> img_out [n,3,8h,8w] fp32 NCHW. */
int sdmi_decode_latent(sdmi_ctx* ctx, const float* latent, int32_t n, float* img_out);

/* Autoencoder::encode_image (src/model/autoencoder/mod.rs:60-66): img [n,3,8h,8w]
 * fp32 NCHW -> latent_out [n,4,h,w] (Encoder::forward, quant_conv, first 4 channels = the
 * posterior mean; the reference does not sample).  Not on the txt2img path;
 * needs the optional weight group autoencoder/encoder/..., autoencoder/quant_conv
 * (SDMI_ERR_STATE otherwise).  Autoencoder::forward (:56-58) = decode_latent(encode_image(x)). */
int sdmi_encode_image(sdmi_ctx* ctx, const float* img, int32_t n, float* latent_out);

/* StableDiffusion::latent_to_image (stablediffusion/mod.rs:69-100):
 * latent [n,4,h,w] -> rgb_out n x [8h,8w,3] uint8 (HWC, truncating cast). */
int sdmi_latent_to_image(sdmi_ctx* ctx, const float* latent, int32_t n, uint8_t* rgb_out);

/* StableDiffusion::sample_image (stablediffusion/mod.rs:51-67). */
int sdmi_sample_image(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T,
                      const float* uncond, int32_t Tu, double scale, size_t n_steps,
                      const float* init_latent, uint64_t seed, uint8_t* rgb_out);

/* qkv_attention (src/model/attention.rs:5-45 == src/backend.rs:88-128; the
 * operator seam of the commented-out `trait Backend`, backend.rs:4-84).
 * q [n,nq,n_state], k,v [n,nk,n_state], mask [>=nq, mask_ld>=nk] additive or
 * NULL -> out [n,nq,n_state]. */
int sdmi_qkv_attention(sdmi_ctx* ctx, const float* q, const float* k, const float* v,
                       const float* mask, int32_t mask_ld, int32_t n, int32_t nq, int32_t nk,
                       int32_t n_state, int32_t n_head, float* out);

/* ---- prompt -> context: tokenizer + CLIP text encoder ---------
 * The step before the hot path.  The CLIP weights (dump subtree clip/...) are an
 * optional group: without them these two functions return SDMI_ERR_STATE and the
 * sampling functions above work unchanged on caller-supplied embeddings. */
typedef struct sdmi_tokenizer sdmi_tokenizer;

/* SimpleTokenizer::new (src/tokenizer.rs:85-120).  The reference opens
 * "bpe_simple_vocab_16e6.txt" in the working directory; here the path is explicit. */
int sdmi_tokenizer_create(sdmi_tokenizer** out, const char* merges_path);
void sdmi_tokenizer_destroy(sdmi_tokenizer* tok);
/* number of vocabulary entries (49408 with the reference's merges file) */
int sdmi_tokenizer_vocab_size(const sdmi_tokenizer* tok);
/* SimpleTokenizer::encode (tokenizer.rs:168-189): UTF-8 text -> ids.  *n_ids is
 * always set to the number of ids the text has; SDMI_ERR_INVALID if capacity is
 * smaller (nothing is written then). */
int sdmi_tokenizer_encode(const sdmi_tokenizer* tok, const char* text, int32_t* ids, int32_t capacity, int32_t* n_ids);
/* SimpleTokenizer::decode (tokenizer.rs:191-196): ids -> UTF-8 (no terminator is
 * written; *n_bytes is always set to the length). */
int sdmi_tokenizer_decode(const sdmi_tokenizer* tok, const int32_t* ids, int32_t n, char* out, int32_t capacity, int32_t* n_bytes);

/* CLIP::forward (src/model/clip/mod.rs:56-75): tokens [n, seq_len] int32 ->
 * out [n, seq_len, ctx_dim]; seq_len <= clip_ctx; causal mask = attn_decoder_mask
 * (src/backend.rs:130-139). */
int sdmi_clip_forward(sdmi_ctx* ctx, const int32_t* tokens, int32_t n, int32_t seq_len, float* out);

/* StableDiffusion::context (src/model/stablediffusion/mod.rs:198-210): tokenises
 * "<|startoftext|>{text}<|endoftext|>" (no padding, no truncation: T = tokens + 2)
 * and runs CLIP.  out holds capacity_tokens x ctx_dim floats; *T is always set;
 * SDMI_ERR_INVALID if T > capacity_tokens or T > clip_ctx.  unconditional_context
 * (:194-196) is context(""), T = 2. */
int sdmi_context(sdmi_ctx* ctx, const sdmi_tokenizer* tok, const char* text, float* out, int32_t capacity_tokens, int32_t* T);

/* save_images (src/bin/sample/main.rs:118-125; image::save_buffer(.., Rgb8)): one 8-bit RGB image
 * [height, width, 3] -> PNG file.  Host code. */
int sdmi_write_png(const char* path, const uint8_t* rgb, int32_t width, int32_t height);

/* ---- hot path, device pointers (zero-copy; same layouts) ---------------------- */
int sdmi_sample_latent_dev(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T,
                           const float* uncond, int32_t Tu, double scale, size_t n_steps,
                           const float* init_latent, float* latent_out);
int sdmi_latent_to_image_dev(sdmi_ctx* ctx, const float* latent, int32_t n, uint8_t* rgb_out);
int sdmi_sample_image_dev(sdmi_ctx* ctx, const float* context, int32_t n, int32_t T,
                          const float* uncond, int32_t Tu, double scale, size_t n_steps,
                          const float* init_latent, uint8_t* rgb_out);

/* ---- multi-GPU: the image batch sharded over the devices of one node ----------------------
 * The reference's caller asks one StableDiffusion for n images of one prompt (src/bin/sample/main.rs:104-109).
 * The path shards over independent images, so a multi-context is: one weights replica + stream + host thread per
 * device inside ONE process, an RCCL communicator over the device list (ncclCommInitAll; librccl is opened
 * lazily here, libsdmi itself links only the HIP runtime), ONE ncclBroadcast over xGMI of the packed prompt
 * embedding [cond | uncond] from the first device per call, contiguous image ranges per device, noise keyed by
 * the GLOBAL image index -- results do not depend on the device count -- and no other collective. */
typedef struct sdmi_multi sdmi_multi;
int sdmi_create_multi(sdmi_multi** out, const sdmi_config* cfg /* .device ignored */, const int32_t* devices, int32_t n_devices);
void sdmi_destroy_multi(sdmi_multi* m);
int32_t sdmi_multi_size(sdmi_multi* m);
/* the per-device context (owned by m): for sdmi_set_weight / sdmi_load_weights_* / sdmi_set_option per device */
sdmi_ctx* sdmi_multi_ctx(sdmi_multi* m, int32_t index);
/* load_stable_diffusion / load_stable_diffusion_model_file on every device in parallel + finalize; kind = "dump" | "burn" */
int sdmi_multi_load_weights(sdmi_multi* m, const char* kind, const char* path);
/* StableDiffusion::sample_image for n_images of ONE prompt: context [T, ctx_dim], uncond [Tu, ctx_dim] (host);
 * init_latents [n_images,4,h,w] or NULL (image i draws N(0,1) from stream seed + i); rgb_out n_images x [8h,8w,3] (host). */
int sdmi_sample_image_sharded(sdmi_multi* m, const float* context, int32_t T, const float* uncond, int32_t Tu,
                              double scale, size_t n_steps, int32_t n_images, const float* init_latents, uint64_t seed,
                              uint8_t* rgb_out);
/* the contiguous global image range [begin, end) device `rank` of `n_ranks` samples (host only): the partition rule of
 * sdmi_sample_image_sharded, and of the one-process-per-GPU launcher (bench.py / sharding.py use the same rule) */
int sdmi_shard_range(int32_t n_images, int32_t rank, int32_t n_ranks, int32_t* begin, int32_t* end);
/* number of RCCL broadcasts issued so far (one per sdmi_sample_image_sharded call) */
int64_t sdmi_multi_broadcast_count(sdmi_multi* m);
/* Diagnostic, needs no device: drives the multi-context's rank runner (one host thread per rank, csrc/multi_ranks.hpp) with n_ranks
 * ranks of which `failing_rank` throws (none if out of range).  Returns SDMI_OK when nothing failed; otherwise the failing rank's status
 * with "rank <r>: injected failure" as the last error -- after verifying that every healthy rank ran to completion and that the
 * drain hook (MultiEngine: synchronise every device's stream) ran exactly once before the error surfaced.  The reference has no
 * multi-device path (src/bin/sample/main.rs:104-109 runs one backend); this pins the error contract of sdmi_sample_image_sharded. */
int sdmi_selftest_rank_errors(int32_t n_ranks, int32_t failing_rank);

/* ---- operator-level entry points (parity tests, profiling) -------------------
 * Same math as the Burn primitives / reference modules named; host pointers
 * in reference layouts.  These let tests/ compare each HIP kernel with the
 * oracle at the op boundary (SURVEY.md section 8c "per-op KATs"). */

/* GroupNorm::forward (+ optional SILU::forward): groupnorm/mod.rs:53-82, silu.rs:14-16.
 * x,out [n,c,h,w]; gamma,beta [c]. */
int sdmi_op_group_norm(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta,
                       int32_t n, int32_t c, int32_t h, int32_t w, int32_t n_group, float eps,
                       int32_t fuse_silu, float* out);
/* precision = 2 only: the same GroupNorm(+SiLU) with its output quantised to MXFP8 (e4m3 elements, one E8M0 scale per 32
 * channels) as the ResBlock convolutions consume it (unet/mod.rs:713-733); `out` is the DEQUANTISED tensor. */
int sdmi_op_group_norm_fp8(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta,
                           int32_t n, int32_t c, int32_t h, int32_t w, int32_t n_group, float eps,
                           int32_t fuse_silu, float* out);
/* Burn nn::LayerNorm (unet/mod.rs:523-525): x,out [rows,c]. */
int sdmi_op_layer_norm(sdmi_ctx* ctx, const float* x, const float* gamma, const float* beta,
                       int32_t rows, int32_t c, float eps, float* out);
/* Burn nn::conv::Conv2d::forward: x [n,cin,h,w], weight [cout,cin,k,k], bias [cout] or NULL,
 * symmetric zero padding `pad`, `stride`; upsample2x != 0 applies the reference's
 * nearest-2x (unet/mod.rs:392-396) to x first.  out [n,cout,ho,wo]. */
int sdmi_op_conv2d(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias,
                   int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t k,
                   int32_t stride, int32_t pad, int32_t upsample2x, float* out);
/* Burn nn::Linear::forward: x [rows,cin] @ weight [cin,cout] + bias. */
int sdmi_op_linear(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias,
                   int32_t rows, int32_t cin, int32_t cout, float* out);
/* GEGLU gate (unet/mod.rs:579-591): proj [rows,2*hidden] -> out [rows,hidden] = a*gelu_erf(gate). */
int sdmi_op_geglu(sdmi_ctx* ctx, const float* proj, int32_t rows, int32_t hidden, float* out);
/* timestep_embedding (unet/mod.rs:19-30): out [dim] for timestep t. */
/* GEGLU::forward (src/model/unet/mod.rs:579-591): x [rows, cin], weight [cin, 2*hidden] (Burn Linear layout), bias
 * [2*hidden] or NULL -> out [rows, hidden] = a * gelu(b), a | b = the halves of x W + bias.  One fused kernel when the
 * shape allows (engine option geglu_fuse), else projection GEMM + gate kernel. */
int sdmi_op_geglu_forward(sdmi_ctx* ctx, const float* x, const float* weight, const float* bias, int32_t rows, int32_t cin,
                          int32_t hidden, float* out);
int sdmi_op_timestep_embedding(sdmi_ctx* ctx, int32_t t, int32_t dim, float* out);

/* ---- tuning / introspection ---------------------------------------------------- */
/* "key=value" option of one context (Engine::set_option), e.g. "profile=1", "gemm_tile=auto".  No reference counterpart (the
 * reference's knobs are Burn backend type parameters, sample/main.rs:59-83).  An unknown key returns SDMI_ERR_INVALID.  The
 * defaults are the measured configuration; the option table -- arithmetic selectors (gemm_f32s, attn_split, gemm_planes and
 * the fp32 semantics of the split kernels at the edges of the range), the precision >= 1 / = 2 selectors, launch geometry,
 * profiling and dump switches -- is DESIGN.md section 11, one line per key. */
int sdmi_set_option(sdmi_ctx* ctx, const char* key, const char* value);
/* time (ms, HIP events on the context stream) and kernel count of the last
 * hot-path call; flops = algorithmic FLOPs it executed (2*MAC of conv/GEMM/attention). */
int sdmi_last_call_stats(sdmi_ctx* ctx, double* gpu_ms, int64_t* n_kernels, double* flops);
/* Per-kernel-class timing collected while option "profile" = "1": HIP events around every
 * launch on the context stream.  cls: 0 conv_gemm (implicit-GEMM conv/linear), 1 splitk_reduce,
 * 2 attention, 3 group_norm(+silu), 4 layer_norm, 5 conv_gemm_fp8 (the MXFP8 convs of precision = 2),
 * 6 conv_gemm_split (precision = 0: the conv/linear launches that run on the bf16 matrix pipe with three-way split fp32
 * operands, k_gemm3x.hip / k_gemm3p.hip; class 0 then holds the launches left on the fp32 matrix instruction), 7 split_rows (fp32 tensors
 * converted to bf16 planes for a plane GEMM outside their producer), 8 other (every launch of the path that is in no other class: layout converters, the
 * CFG + DDIM update, timestep embedding, SiLU of the embedding, row softmax / transposes of the unfused VAE attention, u8 conversion), 9 geglu (the GEGLU gate kernels
 * where the gate is not fused into its GEMM; the quantising gate of precision = 2 included).  flops / bytes are the ALGORITHMIC work of
 * those launches (2*M*N*K; one read + one write of the tensor).  "profile_reset" clears. */
int sdmi_profile_stats(sdmi_ctx* ctx, int32_t cls, double* ms, int64_t* launches, double* flops, double* bytes);
/* What an empty HIP-event pair reads on the context stream (ms): calibrated when "profile" is switched on and already subtracted from
 * every sample of sdmi_profile_stats, so that the classes add up to kernel time and not to kernel time + two event records per launch.
 * (No reference counterpart: the reference has no profiler; bench.py reports it next to the roofline block.) */
int sdmi_profile_overhead(sdmi_ctx* ctx, double* ms);
/* micro-benchmark one implicit-GEMM conv shape on device-resident synthetic
 * data: returns average kernel ms over `iters` launches (HIP events). */
int sdmi_bench_conv(sdmi_ctx* ctx, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                    int32_t k, int32_t stride, int32_t upsample2x, int32_t tile_cfg, int32_t splitk,
                    int32_t iters, double* ms_out);

/* micro-benchmark qkv_attention on device-resident synthetic q/k/v [n, nq|nk, n_state]:
 * average kernel ms over `iters` launches (HIP events). */
int sdmi_bench_attention(sdmi_ctx* ctx, int32_t n, int32_t nq, int32_t nk, int32_t n_state, int32_t n_head,
                         int32_t iters, double* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H */
