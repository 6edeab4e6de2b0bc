// k_split3.hpp -- the exact three-way bf16 split of fp32 values (x = h + m + l), exposed one VALU instruction at a time so that a GEMM
// kernel can place the steps between its matrix instructions itself.  Used by k_gemm3x.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace sdmi {

typedef float s3_f32x4 __attribute__((ext_vector_type(4)));
typedef float s3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int s3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned s3_cvt_pk(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(s3_f32x2{a, b}, s3_bf16x2));
}

// ---- the producer-side split (planes written once, read by k_gemm3p.hip) ---------------------------------------------------------
// Position of slice channel j (0..31) in a 32-element plane row: chunk g = (j & 15) >> 2 holds channels 4g..4g+3 in its elements 0..3
// and channels 16+4g..16+4g+3 in its elements 4..7 -- the k order of the weight planes (launch_pack_split3).
__host__ __device__ constexpr int s3_plane_pos(int j) { return ((j & 15) >> 2) * 8 + ((j >> 4) << 2) + (j & 3); }
// byte offset of channel c (a multiple of 4) of plane pl inside a pixel's planes [C / 32][3][32] bf16
__host__ __device__ constexpr long long s3_plane_byte(int c, int pl) { return (long long)(c >> 5) * 192 + pl * 64 + s3_plane_pos(c & 31) * 2; }

// x -> (h, m, l) bf16 bit patterns with x = h + m + l exactly, and fp32 semantics at the edges of the range (unlike the in-loop
// S3SplitT, which assumes finite operands well inside it): NaN / +-inf stay in h with m = l = 0 (the operand itself is represented as fp32 has it; the PRODUCT inf * w may
// still come out NaN where fp32 gives +-inf, because w's low-order planes can be zero: sdmi.h, "fp32 semantics"), and a finite x that round-to-nearest would carry to inf (|x| >= 0x7F7F8000) takes the TRUNCATED h, which is finite and
// leaves r = x - h exact.  |x| < 2^-109: m / l fall below bf16's normal range and are flushed, i.e. the tail of a tiny value is dropped
// (absolute error < 2^-118 per product operand: sdmi.h, "fp32 semantics").
__device__ __forceinline__ void s3_split1(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    unsigned hp = s3_cvt_pk(x, 0.f) & 0xffffu;
    const bool x_special = (xb & 0x7f800000u) == 0x7f800000u;                  // inf / NaN
    if (!x_special && (hp & 0x7f80u) == 0x7f80u) hp = xb >> 16;               // RNE carried a finite value to inf: truncate instead
    const float r = x_special ? 0.f : x - __builtin_bit_cast(float, hp << 16);
    const unsigned mp = s3_cvt_pk(r, 0.f) & 0xffffu;
    const float r2 = r - __builtin_bit_cast(float, mp << 16);
    h = hp; m = mp; l = s3_cvt_pk(r2, 0.f) & 0xffffu;
}
// four consecutive channels -> 8 bytes per plane
__device__ __forceinline__ void s3_split4(const s3_f32x4 v, unsigned (&h)[2], unsigned (&m)[2], unsigned (&l)[2]) {
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s3_split1(v[i], a[i], b[i], c[i]);
    h[0] = a[0] | (a[1] << 16); h[1] = a[2] | (a[3] << 16);
    m[0] = b[0] | (b[1] << 16); m[1] = b[2] | (b[3] << 16);
    l[0] = c[0] | (c[1] << 16); l[1] = c[2] | (c[3] << 16);
}
// one whole plane chunk: lo = slice channels 4g..4g+3, hi = 16+4g..16+4g+3 -> 16 bytes per plane
__device__ __forceinline__ void s3_split8(const s3_f32x4 lo, const s3_f32x4 hi, s3_u32x4& h, s3_u32x4& m, s3_u32x4& l) {
    unsigned a[2], b[2], c[2];
    s3_split4(lo, a, b, c);
    h[0] = a[0]; h[1] = a[1]; m[0] = b[0]; m[1] = b[1]; l[0] = c[0]; l[1] = c[1];
    s3_split4(hi, a, b, c);
    h[2] = a[0]; h[3] = a[1]; m[2] = b[0]; m[3] = b[1]; l[2] = c[0]; l[3] = c[1];
}
// store the planes of four consecutive channels c..c+3 (c % 4 == 0) of the pixel whose planes start at `pix`
__device__ __forceinline__ void s3_store4(unsigned char* pix, int c, const s3_f32x4 v) {
    unsigned h[2], m[2], l[2];
    s3_split4(v, h, m, l);
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    unsigned char* d = pix + s3_plane_byte(c, 0);
    *reinterpret_cast<u2*>(d) = u2{h[0], h[1]};
    *reinterpret_cast<u2*>(d + 64) = u2{m[0], m[1]};
    *reinterpret_cast<u2*>(d + 128) = u2{l[0], l[1]};
}

// x (8 floats of one lane's fragment) -> three packed-bf16 operands, x = h + m + l exactly.  The 36 instructions are exposed
// one at a time (step s: pair s & 3, phase s >> 2) so that the kernel can place them between matrix instructions itself:
// consecutive steps belong to different pairs, so a dependent instruction is four issue slots behind its producer.
// SCALAR = true: the two residual subtractions of a pair are two v_sub_f32 (inline asm, so that hipcc does not re-pack them)
// instead of one v_pk_add_f32 -- packed fp32 VALU beside MFMAs costs more than its issue slot (MI355X_MICROARCH.md,
// "price of one filler beside MFMAs").
template <bool SCALAR>
struct S3SplitT {
    static constexpr int kSteps = SCALAR ? 44 : 36;
    s3_f32x2 x[4], hf[4], r[4];
    unsigned hp[4], mp[4];
    s3_u32x4 h, m, l;
    __device__ __forceinline__ void load(const s3_f32x4 x0, const s3_f32x4 x1) {
        x[0] = s3_f32x2{x0[0], x0[1]}; x[1] = s3_f32x2{x0[2], x0[3]}; x[2] = s3_f32x2{x1[0], x1[1]}; x[3] = s3_f32x2{x1[2], x1[3]};
    }
    static __device__ __forceinline__ float sub(float a, float b) {
        float d;
        asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    template <int S>
    __device__ __forceinline__ void step() {
        constexpr int i = S & 3, ph = S >> 2;
        if constexpr (!SCALAR) {
            if constexpr (ph == 0) { hp[i] = s3_cvt_pk(x[i][0], x[i][1]); h[i] = hp[i]; }
            else if constexpr (ph == 1) hf[i][0] = __builtin_bit_cast(float, hp[i] << 16);
            else if constexpr (ph == 2) hf[i][1] = __builtin_bit_cast(float, hp[i] & 0xffff0000u);
            else if constexpr (ph == 3) r[i] = x[i] - hf[i];
            else if constexpr (ph == 4) { mp[i] = s3_cvt_pk(r[i][0], r[i][1]); m[i] = mp[i]; }
            else if constexpr (ph == 5) hf[i][0] = __builtin_bit_cast(float, mp[i] << 16);
            else if constexpr (ph == 6) hf[i][1] = __builtin_bit_cast(float, mp[i] & 0xffff0000u);
            else if constexpr (ph == 7) r[i] = r[i] - hf[i];
            else l[i] = s3_cvt_pk(r[i][0], r[i][1]);
        } else {
            if constexpr (ph == 0) { hp[i] = s3_cvt_pk(x[i][0], x[i][1]); h[i] = hp[i]; }
            else if constexpr (ph == 1) hf[i][0] = __builtin_bit_cast(float, hp[i] << 16);
            else if constexpr (ph == 2) hf[i][1] = __builtin_bit_cast(float, hp[i] & 0xffff0000u);
            else if constexpr (ph == 3) r[i][0] = sub(x[i][0], hf[i][0]);
            else if constexpr (ph == 4) r[i][1] = sub(x[i][1], hf[i][1]);
            else if constexpr (ph == 5) { mp[i] = s3_cvt_pk(r[i][0], r[i][1]); m[i] = mp[i]; }
            else if constexpr (ph == 6) hf[i][0] = __builtin_bit_cast(float, mp[i] << 16);
            else if constexpr (ph == 7) hf[i][1] = __builtin_bit_cast(float, mp[i] & 0xffff0000u);
            else if constexpr (ph == 8) r[i][0] = sub(r[i][0], hf[i][0]);
            else if constexpr (ph == 9) r[i][1] = sub(r[i][1], hf[i][1]);
            else l[i] = s3_cvt_pk(r[i][0], r[i][1]);
        }
    }
    template <int S0, int S1>
    __device__ __forceinline__ void steps() {
        if constexpr (S0 < S1) { step<S0>(); steps<S0 + 1, S1>(); }
    }
};

}  // namespace sdmi
