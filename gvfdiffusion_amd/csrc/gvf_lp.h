// gvf_lp.h -- the 16-bit operand type of the matrix pipe as a compile-time trait (internal).
//
// The DiT / VAE kernels contract bf16 OR fp16 operands with fp32 accumulation.  Both are 16-bit storage, so tile layouts, LDS
// images, strides and the weight packers are the same; what differs is (1) the MFMA opcode, (2) the fp32 <-> 16-bit conversions,
// (3) constants (the packed ones of the row-sum trick) and (4) range: fp16 probabilities need a shift (see attn_xt.hip).
// GvfLp<GVF_DT_BF16> / GvfLp<GVF_DT_F16> (dtype codes of include/gvf_dit.h) carry those; kernels take `int DT` as a template argument.
// The reference runs fp16 autocast (inference_dpm_latent.py:122-125); BASELINE.json names bf16: both are first-class.
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(16))) float gvf_f32x16;
typedef __attribute__((ext_vector_type(4))) float gvf_f32x4;

template <int DT>
struct GvfLp;

template <>
struct GvfLp<0> {                                   // bf16: 8 exponent bits (fp32 range), 7 mantissa bits
    typedef __attribute__((ext_vector_type(8))) __bf16 x8;
    typedef __attribute__((ext_vector_type(4))) __bf16 x4;
    static constexpr unsigned ONE2 = 0x3f803f80u;  // (1.0, 1.0) packed
    static constexpr bool kNeedsShift = false;     // exp2 of a raw log2-domain score stays in range
    __device__ static __forceinline__ unsigned pack(float lo, float hi) {      // round-to-nearest-even; hipcc selects v_cvt_pk_bf16_f32
        typedef __attribute__((ext_vector_type(2))) __bf16 x2;
        x2 v;
        v[0] = (__bf16)lo;
        v[1] = (__bf16)hi;
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ static __forceinline__ unsigned short to16(float f) { return (unsigned short)(pack(f, 0.f) & 0xffffu); }
    __device__ static __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
    __device__ static __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
    __device__ static __forceinline__ float from16(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
    __device__ static __forceinline__ gvf_f32x16 mfma32(x8 a, x8 b, gvf_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ gvf_f32x4 mfma16(x8 a, x8 b, gvf_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ gvf_f32x4 mfma4(x4 a, x4 b, gvf_f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0); }
};

template <>
struct GvfLp<1> {                                   // fp16: 5 exponent bits (max 65504, normals from 2^-14), 10 mantissa bits
    typedef __attribute__((ext_vector_type(8))) _Float16 x8;
    typedef __attribute__((ext_vector_type(4))) _Float16 x4;
    static constexpr unsigned ONE2 = 0x3c003c00u;
    static constexpr bool kNeedsShift = true;
    __device__ static __forceinline__ unsigned pack(float lo, float hi) {      // round-to-nearest-even, overflow -> inf: v_cvt_pk_f16_f32
        typedef __attribute__((ext_vector_type(2))) _Float16 x2;
        x2 v;
        v[0] = (_Float16)lo;
        v[1] = (_Float16)hi;
        return __builtin_bit_cast(unsigned, v);
    }
    __device__ static __forceinline__ unsigned short to16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    __device__ static __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    __device__ static __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    __device__ static __forceinline__ float from16(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
    __device__ static __forceinline__ gvf_f32x16 mfma32(x8 a, x8 b, gvf_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ gvf_f32x4 mfma16(x8 a, x8 b, gvf_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ gvf_f32x4 mfma4(x4 a, x4 b, gvf_f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
};

// host side: dispatch a runtime dtype code to a template argument
#define GVF_LP_DISPATCH(dtype_, ...)                          \
    do {                                                      \
        if ((dtype_) == 0) { constexpr int DT = 0; __VA_ARGS__; } \
        else { constexpr int DT = 1; __VA_ARGS__; }           \
    } while (0)
