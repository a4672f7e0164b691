// Shared device/host helpers for libsemipd_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// Opt a kernel in to more than 64 KB of dynamic LDS, once per DEVICE and call site (`done` = one bit per device: the
// attribute is per device, a second GPU in the process would otherwise fail at launch).  Returns non-zero and sets
// the error text when the runtime refuses.
// Compute units this process may use (its HSA_CU_MASK share), as told through semipd_stream_linear_set_cus /
// semipd_gemm_tall_set_cus by the model runner; 0 = never told (whole device assumed by the kernels that ask).
inline std::atomic<int>& owned_cus() {
  static std::atomic<int> v{0};
  return v;
}

inline int ensure_dynamic_lds(const void* kernel, size_t bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed on device %d: %s", what, bytes, dev,
              hipGetErrorString(e));
    return 1;
  }
  done.fetch_or(bit, std::memory_order_relaxed);
  return 0;
}

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

// ---- fp8 KV-cache storage (OCP e5m2 / e4m3fn, gfx950 conversion instructions) -------------------
// The reference stores the cache as `cache_k.to(torch.float8_e5m2)` and the attention kernels load it
// with `.to(q.dtype)` (mem_cache/memory_pool.py:205-209, 326-336): round-to-nearest-even on the way
// in, exact on the way out, all arithmetic in the activation type.
struct f8e5m2_t { uint8_t v; };
struct f8e4m3_t { uint8_t v; };
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <typename F8> struct F8Cvt;
template <> struct F8Cvt<f8e5m2_t> {
  template <bool HI> __device__ static inline uint32_t pack2(float a, float b, uint32_t old) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, HI);
  }
  template <bool HI> __device__ static inline f32x2_t unpack2(uint32_t w) {
    return __builtin_amdgcn_cvt_pk_f32_bf8((int)w, HI);
  }
};
template <> struct F8Cvt<f8e4m3_t> {
  template <bool HI> __device__ static inline uint32_t pack2(float a, float b, uint32_t old) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HI);
  }
  template <bool HI> __device__ static inline f32x2_t unpack2(uint32_t w) {
    return __builtin_amdgcn_cvt_pk_f32_fp8((int)w, HI);
  }
};

// 8 sixteen-bit elements (one 16-byte vector) <-> 8 fp8 bytes
template <typename F8, typename T> __device__ inline uint2 to_f8x8(const Vec16<T>& a) {
  static_assert(Elem<T>::kVec == 8, "fp8 KV storage needs 16-bit activations");
  uint2 w;
  w.x = F8Cvt<F8>::template pack2<false>(Elem<T>::to_f(a.e[0]), Elem<T>::to_f(a.e[1]), 0u);
  w.x = F8Cvt<F8>::template pack2<true>(Elem<T>::to_f(a.e[2]), Elem<T>::to_f(a.e[3]), w.x);
  w.y = F8Cvt<F8>::template pack2<false>(Elem<T>::to_f(a.e[4]), Elem<T>::to_f(a.e[5]), 0u);
  w.y = F8Cvt<F8>::template pack2<true>(Elem<T>::to_f(a.e[6]), Elem<T>::to_f(a.e[7]), w.y);
  return w;
}
template <typename F8, typename T> __device__ inline uint4 from_f8x8(uint2 w) {
  static_assert(Elem<T>::kVec == 8, "fp8 KV storage needs 16-bit activations");
  const f32x2_t a = F8Cvt<F8>::template unpack2<false>(w.x), b = F8Cvt<F8>::template unpack2<true>(w.x);
  const f32x2_t c = F8Cvt<F8>::template unpack2<false>(w.y), d = F8Cvt<F8>::template unpack2<true>(w.y);
  uint4 o;
  o.x = (uint32_t)Elem<T>::from_f(a.x).v | ((uint32_t)Elem<T>::from_f(a.y).v << 16);
  o.y = (uint32_t)Elem<T>::from_f(b.x).v | ((uint32_t)Elem<T>::from_f(b.y).v << 16);
  o.z = (uint32_t)Elem<T>::from_f(c.x).v | ((uint32_t)Elem<T>::from_f(c.y).v << 16);
  o.w = (uint32_t)Elem<T>::from_f(d.x).v | ((uint32_t)Elem<T>::from_f(d.y).v << 16);
  return o;
}

template <typename F8, typename T> __device__ inline F8 to_f8(T x) {
  F8 r;
  r.v = (uint8_t)(F8Cvt<F8>::template pack2<false>(Elem<T>::to_f(x), 0.f, 0u) & 0xffu);
  return r;
}

// KV-pool storage traits: KV == T (rows in the activation type) or an fp8 type
template <typename T, typename KV> struct KVTraits {          // fp8 storage
  static constexpr bool kF8 = true;
  using Raw = uint2;                                           // 8 elements in flight
  __device__ static inline Raw load8(const KV* p) { return *reinterpret_cast<const uint2*>(p); }
  __device__ static inline Raw zero() { return make_uint2(0u, 0u); }
  __device__ static inline uint4 expand(Raw r) { return from_f8x8<KV, T>(r); }
  __device__ static inline void store8(KV* p, const Vec16<T>& a) { *reinterpret_cast<uint2*>(p) = to_f8x8<KV, T>(a); }
  __device__ static inline void store1(KV* p, T x) { *p = to_f8<KV, T>(x); }
};
template <typename T> struct KVTraits<T, T> {                 // same type: byte-exact
  static constexpr bool kF8 = false;
  using Raw = uint4;
  __device__ static inline Raw load8(const T* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static inline Raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static inline uint4 expand(Raw r) { return r; }
  __device__ static inline void store8(T* p, const Vec16<T>& a) { *reinterpret_cast<Vec16<T>*>(p) = a; }
  __device__ static inline void store1(T* p, T x) { *p = x; }
};

// One rotary pair with its roundings pinned: the products with the sine are rounded on their own, the products with the
// cosine are fused into the sums.  Left to the compiler, `x1 * c - x2 * s` is contracted one way or the other depending on
// the surrounding code, and kernels that must write the same bits (the fused decode-step forms against the launches they
// replace) then disagree in the last place of a few outputs.  Every rotation of the library goes through here.
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
  float p = x2 * s, q = x1 * s;
  asm volatile("" : "+v"(p), "+v"(q));
  o1 = __builtin_fmaf(x1, c, -p);
  o2 = __builtin_fmaf(x2, c, q);
  // fp32 VALUES: with an f16 destination the compiler may otherwise fold the fma and the conversion into one
  // v_fma_mixlo_f16 (a single rounding) in one kernel and not in the other
  asm volatile("" : "+v"(o1), "+v"(o2));
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

// Sum of the fp32 K-slice planes of a split GEMM (csrc/stream_linear.hip, gemm8p.hip) at N float4 positions, IN SLICE ORDER
// (the bits of splitk_planes_reduce), with the loads of up to four planes x N positions in flight at once: written as
// `for z: acc += load(z)` the compiler waits for every plane before it asks for the next one (one full memory latency per
// plane: the 5-6 us of the decode step's small kernels were mostly that).
template <int N>
__device__ __forceinline__ void planes_sum_f4(const float* const (&p)[N], int n_planes, int64_t plane_elems, float4 (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = *reinterpret_cast<const float4*>(p[i]);
  int z = 1;
  constexpr int CH = N <= 4 ? 4 : (N <= 8 ? 2 : 1);   // planes per batch: bounded by registers (CH x N float4)
  for (; z + CH <= n_planes; z += CH) {
    float4 t[CH][N];
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int i = 0; i < N; ++i) t[u][i] = *reinterpret_cast<const float4*>(p[i] + (int64_t)(z + u) * plane_elems);
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int i = 0; i < N; ++i) {
        acc[i].x += t[u][i].x; acc[i].y += t[u][i].y; acc[i].z += t[u][i].z; acc[i].w += t[u][i].w;
      }
  }
  if (CH > 1 && z < n_planes) {            // the tail: up to CH - 1 planes, again all loads first
    float4 t[CH > 1 ? CH - 1 : 1][N];
#pragma unroll
    for (int u = 0; u < CH - 1; ++u)
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (z + u < n_planes) t[u][i] = *reinterpret_cast<const float4*>(p[i] + (int64_t)(z + u) * plane_elems);
#pragma unroll
    for (int u = 0; u < CH - 1; ++u)
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (z + u < n_planes) {
          acc[i].x += t[u][i].x; acc[i].y += t[u][i].y; acc[i].z += t[u][i].z; acc[i].w += t[u][i].w;
        }
  }
}
// 8 consecutive floats (two float4) of one row
__device__ __forceinline__ void planes_sum8(const float* p, int n_planes, int64_t plane_elems, float (&f)[8]) {
  const float* const pp[2] = {p, p + 4};
  float4 a[2];
  planes_sum_f4<2>(pp, n_planes, plane_elems, a);
  f[0] = a[0].x; f[1] = a[0].y; f[2] = a[0].z; f[3] = a[0].w; f[4] = a[1].x; f[5] = a[1].y; f[6] = a[1].z; f[7] = a[1].w;
}
// two 8-float runs at once (the two halves of a rotary pair)
__device__ __forceinline__ void planes_sum8x2(const float* pa, const float* pb, int n_planes, int64_t plane_elems, float (&fa)[8],
                                              float (&fb)[8]) {
  const float* const pp[4] = {pa, pa + 4, pb, pb + 4};
  float4 a[4];
  planes_sum_f4<4>(pp, n_planes, plane_elems, a);
  fa[0] = a[0].x; fa[1] = a[0].y; fa[2] = a[0].z; fa[3] = a[0].w; fa[4] = a[1].x; fa[5] = a[1].y; fa[6] = a[1].z; fa[7] = a[1].w;
  fb[0] = a[2].x; fb[1] = a[2].y; fb[2] = a[2].z; fb[3] = a[2].w; fb[4] = a[3].x; fb[5] = a[3].y; fb[6] = a[3].z; fb[7] = a[3].w;
}

}  // namespace semipd
