// Shared device/host helpers for libsemipd_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/semipd.h"

namespace semipd {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define SEMIPD_CHECK_ARG(cond, code, ...)  \
  do {                                      \
    if (!(cond)) {                          \
      ::semipd::set_error(__VA_ARGS__);     \
      return (code);                        \
    }                                       \
  } while (0)

#define SEMIPD_HIP(expr)                                                        \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) {                                                     \
      ::semipd::set_error("%s failed: %s", #expr, hipGetErrorString(_e));       \
      (void)hipGetLastError(); /* do not leave a sticky error for the caller's next launch */ \
      return (int)_e;                                                           \
    }                                                                           \
  } while (0)

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- element type traits -------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kVec = 4;  // elements per 16-byte vector
  __device__ static inline float to_f(float x) { return x; }
  __device__ static inline float from_f(float x) { return x; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  __device__ static inline float to_f(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
  __device__ static inline bf16_t from_f(float f) {  // round-to-nearest-even: v_cvt_pk_bf16_f32
    const __bf16 h = (__bf16)f;
    bf16_t r;
    r.v = __builtin_bit_cast(uint16_t, h);
    return r;
  }
};
template <> struct Elem<f16_t> {
  static constexpr int kVec = 8;
  __device__ static inline float to_f(f16_t x) {
    __half h = __ushort_as_half(x.v);
    return __half2float(h);
  }
  __device__ static inline f16_t from_f(float f) {
    f16_t r;
    r.v = __half_as_ushort(__float2half_rn(f));
    return r;
  }
};

// 16-byte vector of T
template <typename T> struct alignas(16) Vec16 { T e[Elem<T>::kVec]; };

template <typename T> __device__ inline Vec16<T> load16(const T* p) {
  return *reinterpret_cast<const Vec16<T>*>(p);
}
template <typename T> __device__ inline void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<Vec16<T>*>(p) = v;
}

// ---- wave / block reductions ----------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum; every thread gets the result. red must hold >= 16 floats.
__device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

#define SEMIPD_DISPATCH_DTYPE(dtype, T, ...)               \
  switch (dtype) {                                          \
    case SEMIPD_F32: { using T = float; __VA_ARGS__; break; }            \
    case SEMIPD_F16: { using T = ::semipd::f16_t; __VA_ARGS__; break; }   \
    case SEMIPD_BF16: { using T = ::semipd::bf16_t; __VA_ARGS__; break; } \
    default:                                                \
      ::semipd::set_error("unsupported dtype %d", dtype);   \
      return SEMIPD_EDTYPE;                                 \
  }

#define SEMIPD_DISPATCH_HALF(dtype, T, ...)                \
  switch (dtype) {                                          \
    case SEMIPD_F16: { using T = ::semipd::f16_t; __VA_ARGS__; break; }   \
    case SEMIPD_BF16: { using T = ::semipd::bf16_t; __VA_ARGS__; break; } \
    default:                                                \
      ::semipd::set_error("dtype %d not supported here (bf16/f16 only)", dtype); \
      return SEMIPD_EDTYPE;                                 \
  }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace semipd
