// k_split3.hpp -- the exact three-way bf16 split of fp32 values (x = h + m + l), exposed one VALU instruction at a time so that a GEMM
// kernel can place the steps between its matrix instructions itself.  Shared by k_gemm3x.hip (16x16x32 tiles) and k_gemm3y.hip (32x32x16 tiles).
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
