// bmm_fp8 + input_to_float8 for the MLA weight absorption of a block-fp8 DeepSeek model.
//
// The reference multiplies q_nope by W_kc and the attention output by W_vc with both operands in fp8 and one
// fp32 scale per TENSOR: activations are quantised on the fly (input_to_float8, layers/quantization/fp8_utils.py
// :137-149: scale = fp8_max / amax over the whole tensor), the weights once at load time
// (block_quant_to_tensor_quant, fp8_utils.py:152-188), and sgl-kernel's bmm_fp8 (cuBLASLt there,
// sgl-kernel/csrc/gemm/bmm_fp8.cu, python/sgl_kernel/gemm.py:66-82) returns A . B * a_scale * b_scale in bf16.
//
// Here: out^T tiles on the fp8 matrix cores (v_mfma_f32_16x16x32_{fp8,bf8}_{fp8,bf8}), weights as the A operand
// (16 rows of B^T = 16 output columns per wave, 8-byte fragments straight from memory: W_kc / W_vc are 64 KB per
// head and stay in L2), up to four 16-row tiles of the activation as the B operand, fp32 accumulation, one
// multiply by a_scale * b_scale, output through arbitrary (batch, row) strides so that the result lands directly in
// its consumer's layout (q_input[T, H, 576], the o_proj input [T, H * 128]) without a transposing copy.
// gfx950 fp8 is OCP (e4m3fn max 448, e5m2): the reference's HIP branch (e4m3fnuz, 224) is for MI300.
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace semipd {

typedef float bm_f32x4 __attribute__((ext_vector_type(4)));

template <bool A_E5, bool B_E5>
__device__ inline bm_f32x4 bm_mma(long a, long b, bm_f32x4 c) {
  if constexpr (!A_E5 && !B_E5) return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
  else if constexpr (!A_E5 && B_E5) return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a, b, c, 0, 0, 0);
  else if constexpr (A_E5 && !B_E5) return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a, b, c, 0, 0, 0);
}

// out[b, m, n] = sum_k X[b, m, k] * W[b, n, k] * xs * ws; grid (N / 64, ceil(M / (16 MT)), batch), 4 waves.
// W_E5 / X_E5: operand stored as e5m2 instead of e4m3fn.
template <typename OutT, int MT, bool W_E5, bool X_E5>
__global__ void __launch_bounds__(256)
bmm_fp8_kernel(OutT* __restrict__ out, const uint8_t* __restrict__ x, const uint8_t* __restrict__ w,
               const float* __restrict__ x_scale, const float* __restrict__ w_scale, int M, int N, int K,
               int64_t x_bs, int64_t x_rs, int64_t w_bs, int64_t w_ns, int64_t o_bs, int64_t o_rs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int m0 = blockIdx.y * (16 * MT);
  if (n0 >= N) return;
  const uint8_t* wp = w + b * w_bs + (int64_t)min(n0 + c16, N - 1) * w_ns + q4 * 8;
  const uint8_t* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = x + b * x_bs + (int64_t)min(m0 + t * 16 + c16, M - 1) * x_rs + q4 * 8;
  bm_f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = bm_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    const long a = *reinterpret_cast<const long*>(wp + k0);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long bb = *reinterpret_cast<const long*>(xp[t] + k0);
      acc[t] = bm_mma<W_E5, X_E5>(a, bb, acc[t]);
    }
  }
  const float s = x_scale[0] * w_scale[0];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + c16, n = n0 + q4 * 4;
    if (m >= M || n >= N) continue;
    OutT* dst = out + b * o_bs + (int64_t)m * o_rs + n;
    if (n + 4 <= N && (o_rs % 4 == 0) && (o_bs % 4 == 0)) {
      uint2 p;
      p.x = (uint32_t)Elem<OutT>::from_f(acc[t][0] * s).v | ((uint32_t)Elem<OutT>::from_f(acc[t][1] * s).v << 16);
      p.y = (uint32_t)Elem<OutT>::from_f(acc[t][2] * s).v | ((uint32_t)Elem<OutT>::from_f(acc[t][3] * s).v << 16);
      *reinterpret_cast<uint2*>(dst) = p;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) dst[r] = Elem<OutT>::from_f(acc[t][r] * s);
    }
  }
}

// ---- the same product for an UNQUANTISED model's decode batches (bf16 / f16 operands, v_mfma_f32_16x16x32_{bf16,f16}) ----
// torch.bmm(q_nope^T, W_kc) and torch.bmm(attn^T, W_vc) of forward_absorb (models/deepseek_v2.py:655-667, 690-700) are
// [H, <= 64, 128] x [H, 128, 512] and [H, <= 64, 512] x [H, 512, 128]: the library's batched GEMM takes 11-12 us for each
// (profiles/r04_decode_step_deepseek_v2_lite_kernels.txt), all launch and pipeline fill.  Here the weights are kept K-contiguous
// ([H, N, K], like the fp8 buffers above) so that both operands are read as 16-byte MFMA fragments straight from L2.
typedef __bf16 bm_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bm_f16x8 __attribute__((ext_vector_type(8)));
union BmFrag {
  uint4 u;
  bm_bf16x8 b;
  bm_f16x8 f;
};
template <typename T> __device__ inline bm_f32x4 bm_mma16(const BmFrag& a, const BmFrag& b, bm_f32x4 c);
template <> __device__ inline bm_f32x4 bm_mma16<bf16_t>(const BmFrag& a, const BmFrag& b, bm_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
}
template <> __device__ inline bm_f32x4 bm_mma16<f16_t>(const BmFrag& a, const BmFrag& b, bm_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
}

// out[b, m, n] = T(sum_k X[b, m, k] * W[b, n, k]); grid (N / 64, ceil(M / (16 MT)), batch), 4 waves of 16 columns each.
template <typename T, int MT>
__global__ void __launch_bounds__(256)
bmm_nk_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ w, int M, int N, int K, int64_t x_bs,
              int64_t x_rs, int64_t w_bs, int64_t w_ns, int64_t o_bs, int64_t o_rs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int m0 = blockIdx.y * (16 * MT);
  if (n0 >= N) return;
  const T* wp = w + b * w_bs + (int64_t)min(n0 + c16, N - 1) * w_ns + q4 * 8;
  const T* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xp[t] = x + b * x_bs + (int64_t)min(m0 + t * 16 + c16, M - 1) * x_rs + q4 * 8;
  bm_f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = bm_f32x4{0.f, 0.f, 0.f, 0.f};
  // four k-steps (128 k) of fragments in flight before their MFMAs: the loop is a chain of L2 latencies otherwise
  for (int k0 = 0; k0 < K; k0 += 128) {
    BmFrag a[4], bb[4][MT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool ok = k0 + s * 32 < K;
      a[s].u = ok ? *reinterpret_cast<const uint4*>(wp + k0 + s * 32) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t)
        bb[s][t].u = ok ? *reinterpret_cast<const uint4*>(xp[t] + k0 + s * 32) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = bm_mma16<T>(a[s], bb[s][t], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + c16, n = n0 + q4 * 4;
    if (m >= M || n >= N) continue;
    T* dst = out + b * o_bs + (int64_t)m * o_rs + n;
    if (n + 4 <= N && (o_rs % 4 == 0) && (o_bs % 4 == 0)) {
      uint2 p;
      p.x = (uint32_t)Elem<T>::from_f(acc[t][0]).v | ((uint32_t)Elem<T>::from_f(acc[t][1]).v << 16);
      p.y = (uint32_t)Elem<T>::from_f(acc[t][2]).v | ((uint32_t)Elem<T>::from_f(acc[t][3]).v << 16);
      *reinterpret_cast<uint2*>(dst) = p;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) dst[r] = Elem<T>::from_f(acc[t][r]);
    }
  }
}

// ---- input_to_float8: amax over the tensor, then x * (fp8_max / amax) clamped and rounded (RNE) ----
template <typename T>
__global__ void __launch_bounds__(256)
tensor_absmax_kernel(float* __restrict__ block_amax, const T* __restrict__ x, int64_t rows, int K, int M,
                     int64_t x_bs, int64_t x_rs) {
  // rows = batch * M logical rows of K contiguous elements; 8 elements per thread per step.  One partial maximum per
  // block, reduced again by every block of the quantiser: no atomics, nothing to zero between calls (a memset node in
  // front of an atomicMax did not keep its place in a captured decode graph), and the same bits on every replay
  __shared__ float red[16];
  float mx = 0.f;
  const int kv = K / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * kv; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / kv;
    const int c = (int)(i - r * kv);
    const int64_t bb = r / M, m = r - bb * M;
    const Vec16<T> v = load16(x + bb * x_bs + m * x_rs + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(Elem<T>::to_f(v.e[j])));
  }
  mx = wave_max(mx);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m4 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    block_amax[blockIdx.x] = m4;
  }
}

template <typename T, bool E5>
__global__ void __launch_bounds__(256)
tensor_quant_fp8_kernel(uint8_t* __restrict__ q, float* __restrict__ scale_inv, const float* __restrict__ block_amax,
                        int n_partials, const T* __restrict__ x, int64_t rows, int K, int M, int64_t x_bs, int64_t x_rs) {
  const float fp8_max = E5 ? 57344.0f : 448.0f;
  __shared__ float red[4];
  float part = 0.f;
  for (int i = threadIdx.x; i < n_partials; i += 256) part = fmaxf(part, block_amax[i]);
  part = wave_max(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  const float tensor_amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // the reference computes amax, the clamp and the scale as 0-dim tensors of x's dtype (fp8_utils.py:142-147): each
  // is rounded to T, and `fp8_max / amax` with a Python scalar on the left is Tensor.__rtruediv__, i.e.
  // amax.reciprocal() * fp8_max -- two roundings (it differs from the correctly rounded quotient, e.g. for f16 at
  // amax = 300: 1.4941 against 1.4932)
  const float amax = fmaxf(tensor_amax, Elem<T>::to_f(Elem<T>::from_f(1e-12f)));
  const float recip = Elem<T>::to_f(Elem<T>::from_f(1.0f / amax));
  const float scale = Elem<T>::to_f(Elem<T>::from_f(recip * fp8_max));
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_inv[0] = 1.0f / scale;   // scale.float().reciprocal()
  const int kv = K / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * kv; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / kv;
    const int c = (int)(i - r * kv);
    const int64_t bb = r / M, m = r - bb * M;
    const Vec16<T> v = load16(x + bb * x_bs + m * x_rs + c * 8);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // (x * scale) is rounded to T by the reference's tensor arithmetic before the clamp and the cast
      const float p = Elem<T>::to_f(Elem<T>::from_f(Elem<T>::to_f(v.e[j]) * scale));
      y[j] = fminf(fmaxf(p, -fp8_max), fp8_max);
    }
    uint2 p;
    using CV = F8Cvt<typename std::conditional<E5, f8e5m2_t, f8e4m3_t>::type>;
    p.x = CV::template pack2<false>(y[0], y[1], 0u);
    p.x = CV::template pack2<true>(y[2], y[3], p.x);
    p.y = CV::template pack2<false>(y[4], y[5], 0u);
    p.y = CV::template pack2<true>(y[6], y[7], p.y);
    *reinterpret_cast<uint2*>(q + (r * kv + c) * 8) = p;
  }
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_input_to_float8(void* q, float* scale_inv, void* amax_workspace, const void* x, int64_t batch, int64_t m,
                           int64_t k, int64_t x_batch_stride, int64_t x_row_stride, int dtype, int f8_dtype,
                           void* stream) {
  SEMIPD_CHECK_ARG(batch > 0 && m > 0 && k > 0, SEMIPD_EINVAL, "input_to_float8: bad sizes");
  SEMIPD_CHECK_ARG(q && scale_inv && amax_workspace && x, SEMIPD_EINVAL, "input_to_float8: null pointer");
  SEMIPD_CHECK_ARG(f8_dtype == SEMIPD_F8E4M3 || f8_dtype == SEMIPD_F8E5M2, SEMIPD_EDTYPE, "input_to_float8: fp8 type %d", f8_dtype);
  SEMIPD_CHECK_ARG(k % 8 == 0 && x_batch_stride % 8 == 0 && x_row_stride % 8 == 0 && aligned16(x) &&
                   (reinterpret_cast<uintptr_t>(q) & 7u) == 0, SEMIPD_EALIGN, "input_to_float8: k %% 8, 16-byte aligned rows");
  hipStream_t st = as_stream(stream);
  const int64_t rows = batch * m;
  const int64_t items = rows * (k / 8);
  const unsigned grid = (unsigned)std::min<int64_t>((items + 255) / 256, SEMIPD_INPUT_TO_FLOAT8_WORKSPACE_BYTES / 4);
  SEMIPD_DISPATCH_HALF(dtype, T, {
    hipLaunchKernelGGL((tensor_absmax_kernel<T>), dim3(grid), dim3(256), 0, st, (float*)amax_workspace, (const T*)x,
                       rows, (int)k, (int)m, x_batch_stride, x_row_stride);
    if (f8_dtype == SEMIPD_F8E5M2)
      hipLaunchKernelGGL((tensor_quant_fp8_kernel<T, true>), dim3(grid), dim3(256), 0, st, (uint8_t*)q, scale_inv,
                         (const float*)amax_workspace, (int)grid, (const T*)x, rows, (int)k, (int)m, x_batch_stride, x_row_stride);
    else
      hipLaunchKernelGGL((tensor_quant_fp8_kernel<T, false>), dim3(grid), dim3(256), 0, st, (uint8_t*)q, scale_inv,
                         (const float*)amax_workspace, (int)grid, (const T*)x, rows, (int)k, (int)m, x_batch_stride, x_row_stride);
  });
  return launch_status("input_to_float8");
}

int semipd_bmm_nk(void* out, const void* x, const void* w, int64_t batch, int64_t m, int64_t n, int64_t k,
                  int64_t x_batch_stride, int64_t x_row_stride, int64_t w_batch_stride, int64_t w_col_stride,
                  int64_t out_batch_stride, int64_t out_row_stride, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch > 0 && m > 0 && n > 0 && k > 0 && batch <= 65535 && m <= 65535 * 16, SEMIPD_EINVAL, "bmm_nk: bad sizes");
  SEMIPD_CHECK_ARG(out && x && w, SEMIPD_EINVAL, "bmm_nk: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "bmm_nk: bf16 / f16 only");
  SEMIPD_CHECK_ARG(k % 32 == 0 && x_row_stride % 8 == 0 && x_batch_stride % 8 == 0 && w_col_stride % 8 == 0 &&
                   w_batch_stride % 8 == 0 && aligned16(x) && aligned16(w) && (reinterpret_cast<uintptr_t>(out) & 1u) == 0,
                   SEMIPD_EALIGN, "bmm_nk: k %% 32 and 16-byte aligned rows required (X [B, M, K] and W [B, N, K], K contiguous)");
  hipStream_t st = as_stream(stream);
  const int mt = m > 48 ? 4 : (int)((m + 15) / 16);
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 16 * mt - 1) / (16 * mt)), (unsigned)batch);
#define BNK(MTV)                                                                                                        \
  SEMIPD_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((bmm_nk_kernel<T, MTV>), grid, dim3(256), 0, st, (T*)out, (const T*)x, \
                                                    (const T*)w, (int)m, (int)n, (int)k, x_batch_stride, x_row_stride,    \
                                                    w_batch_stride, w_col_stride, out_batch_stride, out_row_stride))
  if (mt == 1) { BNK(1); } else if (mt == 2) { BNK(2); } else if (mt == 3) { BNK(3); } else { BNK(4); }
#undef BNK
  return launch_status("bmm_nk");
}

int semipd_bmm_fp8(void* out, const void* a, const void* b, const float* a_scale, const float* b_scale, int64_t batch,
                   int64_t m, int64_t n, int64_t k, int64_t a_batch_stride, int64_t a_row_stride, int64_t b_batch_stride,
                   int64_t b_col_stride, int64_t out_batch_stride, int64_t out_row_stride, int a_f8_dtype, int b_f8_dtype,
                   int out_dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch > 0 && m > 0 && n > 0 && k > 0 && batch <= 65535, SEMIPD_EINVAL, "bmm_fp8: bad sizes");
  SEMIPD_CHECK_ARG(out && a && b && a_scale && b_scale, SEMIPD_EINVAL, "bmm_fp8: null pointer");
  SEMIPD_CHECK_ARG(k % 32 == 0 && a_row_stride % 8 == 0 && a_batch_stride % 8 == 0 && b_col_stride % 8 == 0 &&
                   b_batch_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(a) & 7u) == 0 &&
                   (reinterpret_cast<uintptr_t>(b) & 7u) == 0, SEMIPD_EALIGN,
                   "bmm_fp8: k %% 32 and 8-byte aligned rows required (A [B, M, K] row-major, B [B, K, N] column-major)");
  const bool a5 = a_f8_dtype == SEMIPD_F8E5M2, b5 = b_f8_dtype == SEMIPD_F8E5M2;
  SEMIPD_CHECK_ARG((a5 || a_f8_dtype == SEMIPD_F8E4M3) && (b5 || b_f8_dtype == SEMIPD_F8E4M3) && !(a5 && b5), SEMIPD_EDTYPE,
                   "bmm_fp8: operands must be e4m3fn / e5m2, not both e5m2 (sgl-kernel/tests/test_bmm_fp8.py:24-25)");
  hipStream_t st = as_stream(stream);
  const int mt = m > 48 ? 4 : (int)((m + 15) / 16);
  dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 16 * mt - 1) / (16 * mt)), (unsigned)batch);
#define BMM_GO(OutT, MTV, W5, X5)                                                                                  \
  hipLaunchKernelGGL((bmm_fp8_kernel<OutT, MTV, W5, X5>), grid, dim3(256), 0, st, (OutT*)out, (const uint8_t*)a,    \
                     (const uint8_t*)b, a_scale, b_scale, (int)m, (int)n, (int)k, a_batch_stride, a_row_stride,     \
                     b_batch_stride, b_col_stride, out_batch_stride, out_row_stride)
#define BMM_MT(OutT, W5, X5)                                                  \
  if (mt == 1) BMM_GO(OutT, 1, W5, X5); else if (mt == 2) BMM_GO(OutT, 2, W5, X5); \
  else if (mt == 3) BMM_GO(OutT, 3, W5, X5); else BMM_GO(OutT, 4, W5, X5)
#define BMM_TYPES(OutT)                                         \
  if (!b5 && !a5) { BMM_MT(OutT, false, false); }               \
  else if (b5) { BMM_MT(OutT, true, false); }                   \
  else { BMM_MT(OutT, false, true); }
  // kernel operand roles: W (the MFMA's A operand) = the caller's B matrix, X = the caller's A matrix
  if (out_dtype == SEMIPD_BF16) { BMM_TYPES(bf16_t) }
  else if (out_dtype == SEMIPD_F16) { BMM_TYPES(f16_t) }
  else { set_error("bmm_fp8: output dtype %d (bf16 / f16 only)", out_dtype); return SEMIPD_EDTYPE; }
#undef BMM_TYPES
#undef BMM_MT
#undef BMM_GO
  return launch_status("bmm_fp8");
}

}  // extern "C"
