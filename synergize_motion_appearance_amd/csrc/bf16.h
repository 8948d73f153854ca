// bf16 storage helpers for the configs[2] ("bf16 storage, fp32 math") kernels: values are kept in HBM as raw 16-bit
// bfloat16 (round-to-nearest-even from fp32: v_cvt_pk_bf16_f32 on gfx950), widened to fp32 in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

typedef float f32x2_t __attribute__((ext_vector_type(2)));
// the pair (a, b) rounded to bf16 in ONE v_cvt_pk_bf16_f32 (a in the low half), as a vector conversion the compiler sees -- pack2's two scalar conversions
// cost an instruction each, and an inline-asm v_cvt_pk_bf16_f32 is invisible to the hazard recognizer: attn_bf3_kernel<false, 2> computed garbage with it
// (tools/scratch/attn_variant.sh), so no split kernel uses the asm form
__device__ __forceinline__ uint32_t cvt2_bf16(float a, float b) {
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// the pair rounded to IEEE half in one v_cvt_pk_f16_f32 (round to nearest even), and a packed pair widened back
__device__ __forceinline__ uint32_t cvt2_f16(float a, float b) {
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_t));
}
__device__ __forceinline__ float f16lo(uint32_t h) { return (float)__builtin_bit_cast(f16x2_t, h)[0]; }
__device__ __forceinline__ float f16hi(uint32_t h) { return (float)__builtin_bit_cast(f16x2_t, h)[1]; }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float a) { return (bf16_t)(pack2(a, 0.f) & 0xffffu); }
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2(a, b), pack2(c, d)); }
__device__ __forceinline__ float4 unpack4(uint2 v) {
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// Storage-type traits for kernels templated on the activation type T (float | bf16_t): 4-element vector access.
template <typename T> struct St;
template <> struct St<float> {
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct St<bf16_t> {
  static __device__ __forceinline__ float4 ld4(const bf16_t* p) { return unpack4(*reinterpret_cast<const uint2*>(p)); }
  static __device__ __forceinline__ void st4(bf16_t* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack4(v.x, v.y, v.z, v.w); }
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
