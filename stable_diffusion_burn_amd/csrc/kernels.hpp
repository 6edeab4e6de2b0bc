// kernels.hpp -- launch interface of the gfx950 HIP kernels of libsdmi.
//
// Layout rules (DESIGN.md "Data layout in HBM"):
//   activations  NHWC fp32 (bf16 for the *_bf16 kernels): [NB][H][W][C]  == row-major [M = NB*H*W][C]
//   conv/linear weights pre-packed "Bt": [N = Cout][K], K ordered
//       k = (cs * T + tap) * CS + ci   with channel c = cs*CS + ci, tap = ky*KW+kx,
//       T = KH*KW, CS = min(32, Cin)   (channel-slice outer, taps inner: the nine
//       taps of one 32-channel slice reuse the same 128-byte pixel segments)
// Every launcher enqueues on `stream` and returns the hipError of the launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdmi {

// ---- implicit-GEMM convolution / linear on fp32 MFMA -------------------------
// C[m][n] = sum_k A(m,k) * Bt[n][k]  (+ bias[n] + rowvec[sample(m)][n] + resid[m][n])
// A(m,k) is gathered from the NHWC source: m -> (nb, oy, ox), k -> (tap, channel).
struct ConvGemm {
    const float* A;       // source activations [NB][Hs][Ws][Cin]
    const float* Bt;      // packed weights [N][K]
    const void* Bt3;      // split kernels (k_gemm3x.hip, k_gemm3p.hip): the same weights as three bf16 planes, [N][K / 32][3][32]
    int b3_grouped;       // ... 0: that row-major layout; 1 (round 5): 16-row fragment groups, [N / 16][K / 32][3][16][32] -- a DMA piece (one plane of one group's k tile) is
                          // 1 KiB of consecutive bytes and a group's k tiles follow each other, instead of sixteen 64-byte pieces 6 K bytes apart (launch_pack_split3)
    const void* A3;       // plane kernel (k_gemm3p.hip): the source activations as three bf16 planes, [NB][Hs][Ws][a3_ld / 192 slices][3][32]
    int a3_ld;            // bytes between source pixels in A3 (192 per 32 channels of the -- possibly wider -- buffer)
    float* C;             // output [M][ldc]; may be null when C3 is set
    void* C3;             // k_gemm3x.hip / k_gemm3p.hip / launch_splitk_reduce: not null -> the output ALSO (or only) as three bf16 planes, [M][ldc3 / 192 slices][3][32]
    int ldc3;             // bytes between output rows in C3 (needs N % 4 == 0: the 16-byte epilogue)
    float* slabs;         // splits > 1: fp32 partial sums [splits][M][N]
    const float* bias;    // [N] or null
    const float* rowvec;  // per-sample per-channel add (time embedding) or null
    const float* resid;   // residual [M][ldr] or null
    int M, N, K;
    int NB, Hs, Ws, Cin;  // source dims (before the optional nearest-2x upsample)
    int Ho, Wo;           // output spatial dims
    int KH, KW, stride, pad, ups;
    int ldc, ldr;
    int a_ld;             // floats between source pixels (normally Cin)
    int b_ld;             // floats between Bt rows (normally K)
    int rowvec_stride;    // floats between samples in rowvec (0: shared by the batch)
    int CS;               // channel slice width: 32, or Cin when Cin < 32
    int kt_total;         // number of 32-wide k tiles
    int kt_per_split;     // k tiles per blockIdx.z slice
    int splits;           // gridDim.z
    long long slab_stride;  // M*N when splits > 1
    unsigned a_bytes, b_bytes;  // extents of A / Bt for the buffer-load range check (v2 kernel)
    int out_mode;               // 0: output in the kernel's storage type; 1: force fp32 (bf16 kernel); 2: bf16 from the fp32 kernel
    const void* zero_page;      // >= 16 readable zero bytes (large-tile kernels: source of padded / out-of-range lanes)
    const void* a_scale;        // fp8 kernel: E8M0 scales of A, [pixels][a_ld / 32] bytes (a_ld = padded channel count = bytes per pixel)
    const void* b_scale;        // fp8 kernel: E8M0 scales of Bt, [N][b_ld / 32] bytes
    int variant;                // k_gemm3x.hip A/B switches (option gemm3x_variant): bit 0 DMA issued in one block per k tile, 1 scalar residual subtractions,
                                // 2 two LDS stages on the 128-row tiles, 4 s_setprio 1 for waves 4-7; k_gemm_bf16x.hip (option gemm_bf16x_variant): bit 0 persistent tile loop
    int resid_acc;              // large-tile bf16 / MXFP8 kernels (round 6): the (bf16) residual is loaded INTO THE ACCUMULATORS in front of the k loop (C = R + A B) instead of
                                // being read by the epilogue one fragment group at a time -- eight exposed load latencies per 256-row tile there; set by Engine::launch_gemm / launch_fp8
    int geglu;                  // large-tile kernels: Bt holds 2 N rows (N value rows, then N gate rows; bias likewise) and the
                                // epilogue writes value * gelu_erf(gate) -- GEGLU::forward (unet/mod.rs:579-591) without the [M, 2N] tensor
    unsigned long long* probe;  // diagnostic (option gemm_probe; k_gemm3p.hip tiles 300 / 303 / 304 only): when non-null the PROBE instantiation runs and
                                // stores 24 words per workgroup (see conv_gemm3p_kernel)
};

// capacity of ConvGemm::probe in workgroups (24 words each); launch_conv_gemm3p refuses a probe launch with a larger grid
constexpr int kGemmProbeBlocks = 1 << 13;

// launch grid of a GEMM kernel: (tiles rounded up to the 8 XCDs) x slices
inline dim3 gemm_grid(const ConvGemm& p, int tiles) {
    return dim3((unsigned)(((tiles + 7) / 8) * 8), 1, (unsigned)p.splits);
}

// tile configurations (index = tile_cfg); BM x BN per 256-thread workgroup
struct GemmTileInfo { int bm, bn; const char* name; };
constexpr int kNumGemmTiles = 10;
const GemmTileInfo& gemm_tile_info(int cfg);
hipError_t launch_conv_gemm2(const ConvGemm& p, int tile_cfg, hipStream_t stream);  // k_gemm2.hip
size_t gemm2_tile_lds_bytes(int cfg);
// bf16 storage / fp32 accumulate (k_gemm_bf16.hip); A, Bt, resid and (unless out_mode == 1) C are bf16
hipError_t launch_conv_gemm_bf16(const ConvGemm& p, int tile_cfg, hipStream_t stream);
hipError_t launch_splitk_reduce_bf16(const ConvGemm& p, hipStream_t stream);
// large-tile (256-row, 8-wave, LDS-DMA staged) bf16 kernel (k_gemm_bf16x.hip); its own tile list
constexpr int kNumGemmTilesX = 4;
const GemmTileInfo& gemm_tile_info_x(int cfg);
hipError_t launch_conv_gemm_bf16x(const ConvGemm& p, int tile_cfg, hipStream_t stream);
// the 256 x 320 / 256 x 256 tiles for 3x3 / stride-1 / pad-1 convolutions with the three taps of a kernel row read from one staged activation tile
// (k_gemm_bf16t.hip); bf16 tile_cfg 100 + kNumGemmTilesX + x.  launch_conv_gemm_bf16t fails (hipErrorInvalidValue) unless conv_gemm_bf16t_supported(p).
constexpr int kNumGemmTilesT = 2;
const GemmTileInfo& gemm_tile_info_t(int cfg);
bool conv_gemm_bf16t_supported(const ConvGemm& p);          // needs p.kt_per_split
hipError_t launch_conv_gemm_bf16t(const ConvGemm& p, int tile_cfg, hipStream_t stream);
// bf16 large tiles as one list: 100 + [0, kNumGemmTilesX) = k_gemm_bf16x.hip, then k_gemm_bf16t.hip
constexpr int kNumGemmTilesXB = kNumGemmTilesX + kNumGemmTilesT;
inline const GemmTileInfo& gemm_tile_info_xb(int c) { return c < kNumGemmTilesX ? gemm_tile_info_x(c) : gemm_tile_info_t(c - kNumGemmTilesX); }
inline hipError_t launch_conv_gemm_bf16_large(const ConvGemm& p, int c, hipStream_t stream) {
    return c < kNumGemmTilesX ? launch_conv_gemm_bf16x(p, c, stream) : launch_conv_gemm_bf16t(p, c - kNumGemmTilesX, stream);
}
// the same structure for fp32 storage (k_gemm2x.hip; Cin % 32 == 0, fp32 output); same tile list
hipError_t launch_conv_gemm2x(const ConvGemm& p, int tile_cfg, hipStream_t stream);
// fp32 on the bf16 matrix pipe: operands as exact sums of three bf16 terms, six partial products (k_gemm3x.hip); its own tile
// list; needs p.Bt3 (launch_pack_split3 of the packed fp32 rows, once at load)
constexpr int kNumGemmTilesS = 6;
const GemmTileInfo& gemm_tile_info_s(int cfg);
hipError_t launch_conv_gemm3x(const ConvGemm& p, int tile_cfg, hipStream_t stream);
hipError_t launch_pack_split3(const float* bt, void* w3, long long rows, int K, hipStream_t s, bool grouped = false);   // grouped needs rows % 16 == 0
// the same arithmetic with the ACTIVATIONS as planes too, written once by their producer (k_gemm3p.hip; tile_cfg 300 + x; needs p.A3)
constexpr int kNumGemmTilesP = 9;
const GemmTileInfo& gemm_tile_info_p(int cfg);
hipError_t launch_conv_gemm3p(const ConvGemm& p, int tile_cfg, hipStream_t stream);
// fp32 rows [rows][ld] (c channels, c % 32 == 0) -> planes [rows][ld3_bytes / 192 slices][3][32] bf16 (slices [0, c / 32) written)
hipError_t launch_split3_rows(const float* x, void* y3, long long rows, int c, long long ld, long long ld3_bytes, hipStream_t s);
hipError_t launch_join3_rows(const void* x3, float* y, long long rows, int c, long long ld3_bytes, long long ld, hipStream_t s);   // planes -> fp32 (exact)
// hipFuncAttributeMaxDynamicSharedMemorySize, once per (kernel, device); safe to call from several host threads (multi.cpp)
hipError_t set_max_dynamic_lds(const void* kernel, int bytes);
// MXFP8 (e4m3 + E8M0 block scales) 256-row LDS-DMA kernel on v_mfma_scale_f32_16x16x128_f8f6f4 (k_fp8.hip); its own tile list
constexpr int kNumGemmTilesQ = 3;
const GemmTileInfo& gemm_tile_info_q(int cfg);
hipError_t launch_conv_gemm_fp8x(const ConvGemm& p, int tile_cfg, hipStream_t stream);
// fp32 OIHW -> e4m3 [Cout][Kp] + scales [Cout][Kp / 32], Kp = roundup(Cin, 128) * kh * kw, k = (slice * T + tap) * 128 + ci
hipError_t launch_pack_conv_weight_fp8(const float* w_oihw, void* bt8, void* bs, int cout, int cin, int kh, int kw, hipStream_t s);
// GroupNorm(+SiLU) of a bf16 tensor with MXFP8 output: y8 [n][hw][Cp] e4m3, y_scale [n][hw][Cp / 32], Cp = roundup(c, 128)
struct GnTune;
hipError_t launch_group_norm_fp8(const void* x, void* y8, void* y_scale, const float* gamma, const float* beta, int n, int hw, int c,
                                 int ldx, int n_group, float eps, bool silu, void* partials, hipStream_t stream, const GnTune* t = nullptr);
hipError_t launch_quantize_fp8(const float* x, void* q, void* s, long long rows, int c, hipStream_t stream);   // fp32 [rows][c] -> MXFP8
// precision = 2 beyond the ResBlock convolutions (option fp8_linear; k_fp8.hip): quantising producers and the Linear weight packer
hipError_t launch_quantize_bf16_fp8(const void* x, void* q, void* s, long long rows, int c, int ldx, hipStream_t stream);   // bf16 [rows][ldx] -> MXFP8 (c % 32 == 0)
hipError_t launch_layer_norm_fp8(const void* x, void* y8, void* y_scale, const float* gamma, const float* beta, int rows, int c, float eps, hipStream_t stream);
hipError_t launch_geglu_fp8(const void* proj, void* y8, void* y_scale, long long rows, int hidden, hipStream_t stream);
hipError_t launch_pack_linear_weight_fp8(const float* w_in_out, void* bt8, void* bs, int cin, int cout, hipStream_t s);
hipError_t launch_dequant_fp8(const void* q, const void* s, float* out, long long rows, int c, hipStream_t stream);
hipError_t launch_pack_conv_weight_bf16(const float* w_oihw, void* bt, int cout, int cin, int kh, int kw, hipStream_t s);
hipError_t launch_pack_linear_weight_bf16(const float* w_in_out, void* bt, int cin, int cout, hipStream_t s);
// sums split-K slabs in fixed order and applies the epilogue
hipError_t launch_splitk_reduce(const ConvGemm& p, hipStream_t stream);
// weight packing (done once at load)
hipError_t launch_pack_conv_weight(const float* w_oihw, float* bt, int cout, int cin, int kh, int kw, hipStream_t s);
hipError_t launch_pack_linear_weight(const float* w_in_out, float* bt, int cin, int cout, hipStream_t s);

// ---- flash-style attention on fp32 MFMA ---------------------------------------
struct AttnParams {
    const float* q; const float* k; const float* v; float* o;
    const int* kv_len;      // [n] valid keys per batch entry, or null (= nk)
    const float* mask;      // additive [>=nq][mask_ld] or null
    int mask_ld;
    int n, n_head, nq, nk, d_head;
    int ldq, ldk, ldv, ldo;             // row strides in floats
    long long q_bs, k_bs, v_bs, o_bs;   // batch strides in floats
    float scale;                        // d_head^-0.25 applied to q and to k (attention.rs:15-26)
    int bf16;                           // q/k/v/o are bf16 in HBM (strides in elements)
    int q_log2;                         // the CALLER states that q already carries d_head^-0.5 log2(e) (attn_bf16_q_scale: folded into the query weight at load, or applied by
                                        // the fp32 -> bf16 conversion): launch_attention_bf16 ignores `scale` and refuses the call without it
    void* o3;                           // fp32 kernels: not null -> the output is written as three bf16 planes instead of fp32 ([n * nq][ldo3 / 192 slices][3][32],
    int ldo3;                           // channel = head * d_head + column; bytes between rows), what the out-projection's k_gemm3p.hip launch reads
    // fp32 kernels, no mask (round 5): kv_splits = S > 1 -> blockIdx.z = key slice: workgroup z walks K / V tiles [z T / S, (z + 1) T / S) and writes its
    // UNNORMALISED output rows + (running maximum in log2 units, row sum) instead of o / o3; launch_attention_combine merges the S slices in slice order.  For the
    // batch-1 levels whose (query tile x head) grid leaves most CUs without a workgroup (32 x 32: 128 workgroups, 16 x 16: 64).
    int kv_splits;
    float* part_o;                      // [S][n][nq][n_head * d_head]
    float* part_ml;                     // [S][n][n_head][nq][2]
    int variant;                        // k_attn_bf16.hip (option attn_bf16_variant): bit 0 = 4-wave workgroups, TWO per CU (independent barriers: the two waves of a SIMD drift out of phase)
    int pack_tail;                      // k_attn_split.hip, d = 40: the packed form of the head's last 8 columns (kernel header); 0 = the six-instruction form (A/B, tests)
};
bool attn_supported_head_dim(int d);
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);
// fp32 q/k/v/o on the bf16 matrix pipe, three-way split operands (k_attn_split.hip): d_head 40 / 80, no additive mask
bool attn_split_supported(const AttnParams& p);
hipError_t launch_attention_split(const AttnParams& p, hipStream_t stream);
// merges the key slices of a kv_splits > 1 launch (either fp32 kernel) into p.o / p.o3
hipError_t launch_attention_combine(const AttnParams& p, hipStream_t stream);
// K / V tile (keys per loop iteration) of the fp32 kernel that would run p: the unit kv_splits cuts
int attn_f32_kv_tile(const AttnParams& p);
// bf16 matrix-core kernel (k_attn_bf16.hip): p.bf16 set, no additive mask; q must arrive multiplied by attn_bf16_q_scale(d_head)
// (the engine folds it into the query projection's weight at load) and the caller must say so (p.q_log2, else hipErrorInvalidValue); p.scale is not used
inline float attn_bf16_q_scale(int d_head) { return (float)(1.4426950408889634 / __builtin_sqrt((double)d_head)); }
// the head dims whose bf16 q tensors follow that convention (the fused bf16 kernel's; every other head dim keeps the reference's scale in the kernel)
inline bool attn_bf16_q_is_log2(int d_head) { return d_head == 40 || d_head == 80 || d_head == 160; }
hipError_t launch_attention_bf16(const AttnParams& p, hipStream_t stream);
// row softmax (in place) for the unfused single-head VAE attention: x[rows][cols] *= scale first
hipError_t launch_softmax_rows(float* x, int rows, int cols, float scale, hipStream_t stream);

// ---- normalisation (HBM-bound class) --------------------------------------------
// GroupNorm (+SiLU) over NHWC: stats pass (per-chunk partial sums) + apply pass.
// `partials` needs gn_partials_bytes(n, hw, c) bytes of scratch.
size_t gn_partials_bytes(int n, int hw, int c, int min_wgs = 0);    // min_wgs: k_norm.hip gn_geom (option gn32_min_wgs)
// ldx: elements between pixels of x (>= c; x may be a channel slice of a wider buffer); y is dense [n][hw][c]
hipError_t launch_group_norm(const float* x, float* y, const float* gamma, const float* beta,
                             int n, int hw, int c, int ldx, int n_group, float eps, bool silu,
                             void* partials, hipStream_t stream, int min_wgs = 0);
hipError_t launch_layer_norm(const float* x, float* y, const float* gamma, const float* beta,
                             int rows, int c, float eps, hipStream_t stream);
// the same normalisations with the result written as three bf16 planes (k_split3.hpp; y3 dense: (c / 32) * 192 bytes per pixel / row),
// what the consuming k_gemm3p.hip launch reads.  c % 32 == 0.
hipError_t launch_group_norm_planes(const float* x, void* y3, const float* gamma, const float* beta, int n, int hw, int c, int ldx,
                                    int n_group, float eps, bool silu, void* partials, hipStream_t stream, int min_wgs = 0);
hipError_t launch_layer_norm_planes(const float* x, void* y3, const float* gamma, const float* beta, int rows, int c, float eps,
                                    hipStream_t stream);

// ---- elementwise / data movement ---------------------------------------------------
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, float scale, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, hipStream_t s);
hipError_t launch_concat_channels(const float* a, const float* b, float* dst, long long rows, int ca, int cb, hipStream_t s);
hipError_t launch_geglu(const float* proj, float* out, long long rows, int hidden, hipStream_t s);
hipError_t launch_geglu_planes(const float* proj, void* out3, long long rows, int hidden, hipStream_t s);   // out3: planes, hidden % 32 == 0
hipError_t launch_silu(const float* x, float* y, long long n, hipStream_t s);
hipError_t launch_repeat_rows(const float* src, float* dst, int n, long long elems, hipStream_t s);   // dst[i][:] = src[:], i < n
hipError_t launch_nhwc_to_nchw_slice(const float* src, float* dst, int n, int c_src, int c_out, int h, int w, hipStream_t s);
hipError_t launch_nchw3_to_nhwc4(const float* src, float* dst, int n, int h, int w, hipStream_t s);
// CLIP text encoder pieces (clip/mod.rs): QuickGELU in place, token + position embedding, decoder mask
hipError_t launch_quick_gelu(float* x, long long n, hipStream_t s);
hipError_t launch_clip_embed(const int* tokens, const float* tok_table, const float* pos_table, float* out, int n, int T, int C,
                             hipStream_t s);
hipError_t launch_causal_mask(float* mask, int T, hipStream_t s);
// dst[c][r] = src[r*src_ld + c]
hipError_t launch_transpose2d(const float* src, float* dst, int rows, int cols, int src_ld, hipStream_t s);
// out[s][0:half] = cos(t_s * f_i), out[s][half:dim] = sin(t_s * f_i)  (unet/mod.rs:19-30)
hipError_t launch_timestep_embedding(const int* t_dev, int n_t, int dim, float* out, hipStream_t s);
// CFG combine + DDIM update on NHWC latents (stablediffusion/mod.rs:152-156,190-191).
// eps [2n][hw][4] (uncond rows first), latent [n][hw][4] updated in place, and
// copied twice into unet_in [2n][hw][4] for the next step.
struct DdimCoef { float scale, sqrt_noise, inv_div /*unused*/, sqrt_cur, sqrt_prev, dir_coef; };
hipError_t launch_cfg_ddim(const float* eps, float* latent, float* unet_in, long long per_half,
                           DdimCoef c, hipStream_t s);
hipError_t launch_dup_latent(const float* latent, float* unet_in, long long per_half, hipStream_t s);
// (img+1)/2*255 -> clamp -> truncating u8, NHWC in, HWC out (stablediffusion/mod.rs:79-99)
hipError_t launch_image_to_u8(const float* img_nhwc, uint8_t* out, long long n_elem, hipStream_t s);
hipError_t launch_fill_normal(float* dst, long long n, uint64_t seed, hipStream_t s);

// ---- bf16-storage variants (k_bf16.hip) ---------------------------------------------------------------
// Launch geometry of the bf16 GroupNorm passes (statistics, apply, the MXFP8 apply of k_fp8.hip): a sample's hw rows are cut into `chunks` ranges, one workgroup
// of cq x R threads each (cq = c / 8 columns of 8 channels, R pixel rows per pass).  Round 4 cut by size alone (32 KB per chunk): at the batches of
// BASELINE.json configs[2..4] that is 2 560 workgroups of 2-3 loads per thread for a 64 x 64 x 320 tensor, and every one pays the fixed tail (LDS reduction, the fp64
// merges; the apply pass re-reads all chunk partials).  target_wgs > 0 (round 5) also bounds the chunks per sample by target_wgs / n: one round of larger chunks.
// Defaults measured in round 5 (profiles/r05d_*): GroupNorm class -21 % (bf16 B = 16), -24 % (MXFP8), images/s +2.0 % / +2.6 %; {0, 1024, 1} is round 4's geometry.
struct GnTune { int target_wgs = 512; int max_threads = 512; int unroll = 2; };   // unroll: independent 16-byte loads in flight per thread (1, 2 or 4)
struct GnGeomH { int cq, R, threads, chunks, rows_per_chunk; };
GnGeomH gn_geom_bf16(int n, int hw, int c, GnTune t);
size_t gn_partials_bytes_bf16(int n, int hw, int c, GnTune t = GnTune());
hipError_t launch_group_norm_bf16(const void* x, void* y, const float* gamma, const float* beta, int n, int hw, int c, int ldx,
                                  int n_group, float eps, bool silu, void* partials, hipStream_t stream, GnTune t = GnTune());
// the statistics half of launch_group_norm_bf16 alone (shared with the fp8-output apply of k_fp8.hip)
hipError_t launch_group_norm_bf16_stats(const void* x, int n, int hw, int c, int ldx, int n_group, void* partials, hipStream_t stream, GnTune t = GnTune());
hipError_t launch_layer_norm_bf16(const void* x, void* y, const float* gamma, const float* beta, int rows, int c, float eps,
                                  hipStream_t stream);
hipError_t launch_geglu_bf16(const void* proj, void* out, long long rows, int hidden, hipStream_t s);
hipError_t launch_f32_to_bf16(const float* src, void* dst, long long n, hipStream_t s);
hipError_t launch_f32_to_bf16_scaled(const float* src, void* dst, long long n, float scale, hipStream_t s);   // dst = bf16(src * scale)
hipError_t launch_scale_f32(float* x, long long n, float scale, hipStream_t s);                                // in place
hipError_t launch_nhwc_bf16_to_nchw_f32(const void* src, float* dst, int n, int c, int h, int w, hipStream_t s);
hipError_t launch_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int n, int c, int h, int w, float scale, hipStream_t s);
hipError_t launch_transpose2d_bf16(const void* src, void* dst, int rows, int cols, int src_ld, hipStream_t s);
hipError_t launch_softmax_rows_f32_to_bf16(const float* x, void* y, int rows, int cols, float scale, hipStream_t s);

}  // namespace sdmi
