// MFMA 32x32x16 operand / accumulator types shared by the prefill attention kernels (gfx950).
#pragma once
#include "common.h"

namespace semipd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  typedef bf16x8_t frag;
  __device__ static inline f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  typedef f16x8_t frag;
  __device__ static inline f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

union Frag16 {  // 16 bytes viewed as MFMA operand / raw words / elements
  uint4 u;
  uint32_t w[4];
  s16x4 s[2];
  uint16_t e[8];
  bf16x8_t b;
  f16x8_t f;
};
template <typename T> __device__ inline typename Mfma<T>::frag as_frag(const Frag16& x);
template <> __device__ inline bf16x8_t as_frag<bf16_t>(const Frag16& x) { return x.b; }
template <> __device__ inline f16x8_t as_frag<f16_t>(const Frag16& x) { return x.f; }

// two floats -> one dword of T, round to nearest even: ONE v_cvt_pk_{bf16,f16}_f32 (element-wise conversions and an
// OR cost three instructions per pair; 32 pairs per KV tile in the softmax)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ inline uint32_t pack2(float a, float b);
template <> __device__ inline uint32_t pack2<bf16_t>(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <> __device__ inline uint32_t pack2<f16_t>(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}

}  // namespace semipd
