// MLA (DeepSeek) paged decode attention, stage 1, on the matrix cores (gfx950).
//
// The KV pool holds ONE latent row per token: 576 bf16 = [512 compressed | 64 rope]; keys are the
// whole row, values are its first 512 columns (absorbed formulation, models/deepseek_v2.py:633-706),
// and all heads share it (MQA).  A workgroup owns (request, 16-head tile, kv split); its four waves
// share one 32-token tile of latent rows in LDS, fetched from HBM exactly once per head tile:
//   every wave:   S^T[token, head] = tile x Q^T          36 x v_mfma_f32_16x16x32 (A = ds_read_b128)
//   wave w:       O^T[128w..128w+127, head] += V^T P^T     8 x v_mfma_f32_16x16x32 (A = ds_read_b64_tr_b16)
// The QK^T product is recomputed by each wave (4 x redundant) because the matrix pipes are idle
// anyway: per tile a wave issues 44 MFMAs (~700 cycles) while the tile's 36.9 KB take ~3500 cycles of
// the CU's share of HBM bandwidth.  Softmax bookkeeping is identical in the four waves, so no
// cross-wave exchange is needed; the next tile is prefetched into registers during the MFMAs.
//
// Replaces _fwd_grouped_kernel_stage1 for Lk = 576 / Lv = 512
// (layers/attention/triton_ops/decode_attention.py:234-390, BLOCK_DPE = 64 path).
#include "common.h"

#include <type_traits>

namespace semipd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

union FragM {
  uint4 u;
  uint16_t e[8];
  s16x4 s[2];
  bf16x8_t b;
  f16x8_t f;
};
template <typename T> struct MfmaM;
template <> struct MfmaM<bf16_t> {
  __device__ static inline f32x4 mma(const FragM& a, const FragM& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct MfmaM<f16_t> {
  __device__ static inline f32x4 mma(const FragM& a, const FragM& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
  }
};

// KV = storage type of the latent rows: T, or an OCP fp8 type (--kv-cache-dtype fp8_*, memory_pool.py:439-452): 8
// bytes per 8 elements in flight, expanded to T when the tile is written to LDS.
template <typename T, typename KV>
__global__ void __launch_bounds__(256, 2)
mla_decode_kernel(T* __restrict__ out, const T* __restrict__ q, const KV* __restrict__ kv_buf,
                  const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices,
                  float* __restrict__ attn_logits, int num_q_heads, int tiles, int64_t q_stride,
                  int64_t o_stride, int64_t kvbuf_stride, int num_kv_splits, float sm_scale,
                  float logit_cap) {
  constexpr int DK = 576, DV = 512, KS = DK / 32, TOK = 32;
  constexpr int RS = DK * 2 + 64;  // 1216 B = 304 dwords = 48 (mod 64): conflict-free transposing reads
  constexpr int CPR = DK / 8;      // 72 16-byte chunks per row
  constexpr int NI = TOK * CPR / 256;  // 9 chunks per thread per tile
  constexpr int DTW = DV / 16 / 4;     // 8 d-tiles of 16 per wave
  __shared__ __attribute__((aligned(16))) uint8_t tile[TOK * RS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int split = blockIdx.x % num_kv_splits;
  const int tmp = blockIdx.x / num_kv_splits;
  const int htile = tmp % tiles;
  const int b = tmp / tiles;
  const int h0 = htile * 16;
  const int heads = min(16, num_q_heads - h0);
  const bool head_ok = c16 < heads;

  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  if (s_end <= s_begin) {
    if (num_kv_splits == 1 && head_ok) {
      for (int d = wave * 128 + q4; d < wave * 128 + 128; d += 4)
        out[(int64_t)b * o_stride + (int64_t)(h0 + c16) * DV + d] = Elem<T>::from_f(0.f);
    }
    return;
  }

  // Q^T fragments (B operand): lane = head c16, d = ks*32 + q4*8 .. +8
  FragM qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks].u = make_uint4(0, 0, 0, 0);
    if (head_ok)
      qf[ks].u = *reinterpret_cast<const uint4*>(q + (int64_t)b * q_stride + (int64_t)(h0 + c16) * DK +
                                                 ks * 32 + q4 * 8);
  }
  f32x4 o_acc[DTW];
#pragma unroll
  for (int t = 0; t < DTW; ++t) o_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int32_t* idx_base = kv_indices + kv_start;
  const int n_tiles = (s_end - s_begin + TOK - 1) / TOK;
  using KVT = KVTraits<T, KV>;
  int32_t idx[NI];
  typename KVT::Raw reg[NI];
  auto load_idx = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int tok = s_begin + ti * TOK + (tid + i * 256) / CPR;
      idx[i] = (ti < n_tiles && tok < s_end) ? idx_base[tok] : 0;
    }
  };
  auto fetch = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + i * 256;
      const int r = item / CPR, ch = item - r * CPR;
      reg[i] = KVT::zero();  // rows past the end must be zero (0 * garbage could be NaN)
      if (s_begin + ti * TOK + r < s_end) reg[i] = KVT::load8(kv_buf + (int64_t)idx[i] * kvbuf_stride + ch * 8);
    }
  };

  const uint8_t* krow = tile + c16 * RS + q4 * 16;                           // + t*16*RS + ks*64
  const uint8_t* vrow = tile + (q4 * 4 + (c16 >> 2)) * RS + (c16 & 3) * 8 + wave * 256;  // + dt*32 (+16*RS)

  load_idx(0);
  fetch(0);
  load_idx(1);
  for (int ti = 0; ti < n_tiles; ++ti) {
    __syncthreads();  // every wave is done with the previous tile
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + i * 256;
      const int r = item / CPR, ch = item - r * CPR;
      *reinterpret_cast<uint4*>(tile + r * RS + ch * 16) = KVT::expand(reg[i]);
    }
    __syncthreads();
    if (ti + 1 < n_tiles) fetch(ti + 1);
    load_idx(ti + 2);

    // ---- S^T = tile x Q^T (every wave, all 32 tokens x 16 heads) ----
    f32x4 s_acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FragM a;
        a.u = *reinterpret_cast<const uint4*>(krow + t * 16 * RS + ks * 64);
        s_acc[t] = MfmaM<T>::mma(a, qf[ks], s_acc[t]);
      }
    }
    // ---- online softmax: lane = head c16, tokens t*16 + q4*4 + r ----
    const int base_tok = s_begin + ti * TOK;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float s = s_acc[t][rr] * sm_scale;
        if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
        s = (base_tok + t * 16 + q4 * 4 + rr < s_end) ? s : -INFINITY;
        s_acc[t][rr] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DTW; ++t) o_acc[t] *= alpha;
      m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
    FragM pf;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float p = __expf(s_acc[t][rr] - m_use);
        psum += p;
        pf.e[t * 4 + rr] = Elem<T>::from_f(p).v;
      }
    }
    l_run += psum;
    // ---- this wave's 128 output dims: O^T += V^T P^T ----
#pragma unroll
    for (int dt = 0; dt < DTW; ++dt) {
      FragM a;
      a.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + dt * 32));
      a.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + 16 * RS + dt * 32));
      o_acc[dt] = MfmaM<T>::mma(a, pf, o_acc[dt]);
    }
  }

  // ---- epilogue: wave w writes d in [128w, 128w+128) ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (!head_ok) return;
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  const int hq = h0 + c16;
  if (num_kv_splits == 1) {
    T* orow = out + (int64_t)b * o_stride + (int64_t)hq * DV + wave * 128;
#pragma unroll
    for (int dt = 0; dt < DTW; ++dt) {
      uint2 w;
      w.x = (uint32_t)Elem<T>::from_f(o_acc[dt][0] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][1] * inv).v << 16);
      w.y = (uint32_t)Elem<T>::from_f(o_acc[dt][2] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][3] * inv).v << 16);
      *reinterpret_cast<uint2*>(orow + dt * 16 + q4 * 4) = w;
    }
  } else {
    float* dst = attn_logits + (((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (DV + 1);
#pragma unroll
    for (int dt = 0; dt < DTW; ++dt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dst[wave * 128 + dt * 16 + q4 * 4 + rr] = o_acc[dt][rr] * inv;
    }
    if (wave == 0 && q4 == 0) dst[DV] = m_run + __logf(l_tot);
  }
}

// More than 16 heads per rank (DeepSeek-V3 below TP = 8: up to 128): the kernel above would fetch every latent
// tile once per 16-head tile (0.9 TB/s of algorithmic bytes at 128 heads).  Here a workgroup owns (request, group
// of HT head tiles, kv split) and has 2 * HT waves that share the 32-token tile in LDS: wave (ht, half) computes
// S^T for head tile ht (36 MFMAs, done twice per head tile) and the 256 output dims `half` of it (32 MFMAs).  A
// tile is read once per 16 * HT heads -- twice for 128 heads with HT = 4, the second read served by L2 /
// Infinity Cache.  (One wave per head tile with all 512 dims needs 72 (Q^T) + 128 (O^T) registers + state: it
// spills at two waves per SIMD.)
template <typename T, typename KV, int HT>
__global__ void __launch_bounds__(128 * HT, HT == 4 ? 2 : 1)
mla_decode_wide_kernel(T* __restrict__ out, const T* __restrict__ q, const KV* __restrict__ kv_buf,
                       const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices,
                       float* __restrict__ attn_logits, int num_q_heads, int head_groups, int64_t q_stride,
                       int64_t o_stride, int64_t kvbuf_stride, int num_kv_splits, float sm_scale, float logit_cap) {
  constexpr int DK = 576, DV = 512, KS = DK / 32, TOK = 32;
  constexpr int RS = DK * 2 + 64;
  constexpr int CPR = DK / 8;
  constexpr int NW = 2 * HT;
  constexpr int NT = 64 * NW;
  constexpr int NI = (TOK * CPR + NT - 1) / NT;
  constexpr int DT = DV / 16 / 2;        // 16 d-tiles of 16: this wave's half of the output dims
  __shared__ __attribute__((aligned(16))) uint8_t tile[TOK * RS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int split = blockIdx.x % num_kv_splits;
  const int tmp = blockIdx.x / num_kv_splits;
  const int hgroup = tmp % head_groups;
  const int b = tmp / head_groups;
  const int h0 = (hgroup * HT + (wave >> 1)) * 16;
  const int dhalf = wave & 1;
  const int heads = min(16, max(0, num_q_heads - h0));
  const bool head_ok = c16 < heads;

  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  if (s_end <= s_begin) {
    if (num_kv_splits == 1 && head_ok) {
      for (int d = dhalf * 256 + q4; d < dhalf * 256 + 256; d += 4)
        out[(int64_t)b * o_stride + (int64_t)(h0 + c16) * DV + d] = Elem<T>::from_f(0.f);
    }
    return;
  }

  FragM qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks].u = make_uint4(0, 0, 0, 0);
    if (head_ok)
      qf[ks].u = *reinterpret_cast<const uint4*>(q + (int64_t)b * q_stride + (int64_t)(h0 + c16) * DK + ks * 32 + q4 * 8);
  }
  f32x4 o_acc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int32_t* idx_base = kv_indices + kv_start;
  const int n_tiles = (s_end - s_begin + TOK - 1) / TOK;
  using KVT = KVTraits<T, KV>;
  int32_t idx[NI];
  typename KVT::Raw reg[NI];
  auto load_idx = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + i * NT;
      const int tok = s_begin + ti * TOK + item / CPR;
      idx[i] = (ti < n_tiles && item < TOK * CPR && tok < s_end) ? idx_base[tok] : 0;
    }
  };
  auto fetch = [&](int ti) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + i * NT;
      const int r = item / CPR, ch = item - r * CPR;
      reg[i] = KVT::zero();
      if (item < TOK * CPR && s_begin + ti * TOK + r < s_end)
        reg[i] = KVT::load8(kv_buf + (int64_t)idx[i] * kvbuf_stride + ch * 8);
    }
  };

  const uint8_t* krow = tile + c16 * RS + q4 * 16;
  const uint8_t* vrow = tile + (q4 * 4 + (c16 >> 2)) * RS + (c16 & 3) * 8 + dhalf * 512;

  load_idx(0);
  fetch(0);
  load_idx(1);
  for (int ti = 0; ti < n_tiles; ++ti) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int item = tid + i * NT;
      const int r = item / CPR, ch = item - r * CPR;
      if (item < TOK * CPR) *reinterpret_cast<uint4*>(tile + r * RS + ch * 16) = KVT::expand(reg[i]);
    }
    __syncthreads();
    if (ti + 1 < n_tiles) fetch(ti + 1);
    load_idx(ti + 2);
    if (heads == 0) continue;              // wave-uniform: a padding head tile only helps with the staging

    f32x4 s_acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FragM a;
        a.u = *reinterpret_cast<const uint4*>(krow + t * 16 * RS + ks * 64);
        s_acc[t] = MfmaM<T>::mma(a, qf[ks], s_acc[t]);
      }
    }
    const int base_tok = s_begin + ti * TOK;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float sv = s_acc[t][rr] * sm_scale;
        if (logit_cap > 0.f) sv = logit_cap * tanhf(sv / logit_cap);
        sv = (base_tok + t * 16 + q4 * 4 + rr < s_end) ? sv : -INFINITY;
        s_acc[t][rr] = sv;
        mx = fmaxf(mx, sv);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DT; ++t) o_acc[t] *= alpha;
      m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
    FragM pf;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float p = __expf(s_acc[t][rr] - m_use);
        psum += p;
        pf.e[t * 4 + rr] = Elem<T>::from_f(p).v;
      }
    }
    l_run += psum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      FragM a;
      a.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + dt * 32));
      a.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + 16 * RS + dt * 32));
      o_acc[dt] = MfmaM<T>::mma(a, pf, o_acc[dt]);
    }
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (!head_ok) return;
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  const int hq = h0 + c16;
  if (num_kv_splits == 1) {
    T* orow = out + (int64_t)b * o_stride + (int64_t)hq * DV + dhalf * 256;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint2 w;
      w.x = (uint32_t)Elem<T>::from_f(o_acc[dt][0] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][1] * inv).v << 16);
      w.y = (uint32_t)Elem<T>::from_f(o_acc[dt][2] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][3] * inv).v << 16);
      *reinterpret_cast<uint2*>(orow + dt * 16 + q4 * 4) = w;
    }
  } else {
    float* dst = attn_logits + (((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (DV + 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dst[dhalf * 256 + dt * 16 + q4 * 4 + rr] = o_acc[dt][rr] * inv;
    }
    if (q4 == 0 && dhalf == 0) dst[DV] = m_run + __logf(l_tot);
  }
}

// defined in mla_decode_shared.hip; -1 = shape not covered
template <typename T>
int launch_mla_decode_shared(T* out, const T* q, const T* kv_buf, const int32_t* kv_indptr, const int32_t* kv_indices,
                             float* attn_logits, int64_t batch, int Hq, int64_t q_stride, int64_t o_stride,
                             int64_t kvbuf_stride, int splits, float sm_scale, float logit_cap, hipStream_t st);

template <typename T, typename KV>
int launch_mla_decode(T* out, const T* q, const KV* kv_buf, const int32_t* kv_indptr, const int32_t* kv_indices,
                      float* attn_logits, int64_t batch, int Hq, int64_t q_stride, int64_t o_stride,
                      int64_t kvbuf_stride, int splits, float sm_scale, float logit_cap, hipStream_t st) {
  const int tiles = (Hq + 15) / 16;
  if constexpr (std::is_same<T, KV>::value) {
    // 64 / 128 heads per rank, rows in the activation type: the shared-tile kernel (mla_decode_shared.hip)
    const int rc = launch_mla_decode_shared<T>(out, q, kv_buf, kv_indptr, kv_indices, attn_logits, batch, Hq, q_stride,
                                               o_stride, kvbuf_stride, splits, sm_scale, logit_cap, st);
    if (rc >= 0) return rc;
  }
  if (tiles > 1) {
    // several head tiles per rank: two waves per head tile, the latent tile shared by 2 or 4 head tiles
    const int ht = tiles > 2 ? 4 : 2;
    const int groups = (tiles + ht - 1) / ht;
    const int64_t total_w = batch * groups * splits;
    if (total_w > 0x7fffffff) {
      set_error("mla_decode: grid too large");
      return SEMIPD_EINVAL;
    }
    if (ht == 4)
      hipLaunchKernelGGL((mla_decode_wide_kernel<T, KV, 4>), dim3((unsigned)total_w), dim3(512), 0, st, out, q, kv_buf,
                         kv_indptr, kv_indices, attn_logits, Hq, groups, q_stride, o_stride, kvbuf_stride, splits,
                         sm_scale, logit_cap);
    else
      hipLaunchKernelGGL((mla_decode_wide_kernel<T, KV, 2>), dim3((unsigned)total_w), dim3(256), 0, st, out, q, kv_buf,
                         kv_indptr, kv_indices, attn_logits, Hq, groups, q_stride, o_stride, kvbuf_stride, splits,
                         sm_scale, logit_cap);
    return launch_status("mla_decode_wide");
  }
  const int64_t total = batch * tiles * splits;
  if (total > 0x7fffffff) {
    set_error("mla_decode: grid too large");
    return SEMIPD_EINVAL;
  }
  hipLaunchKernelGGL((mla_decode_kernel<T, KV>), dim3((unsigned)total), dim3(256), 0, st, out, q, kv_buf, kv_indptr,
                     kv_indices, attn_logits, Hq, tiles, q_stride, o_stride, kvbuf_stride, splits, sm_scale,
                     logit_cap);
  return launch_status("mla_decode");
}

#define MLA_INST(T, KV)                                                                                              \
  template int launch_mla_decode<T, KV>(T*, const T*, const KV*, const int32_t*, const int32_t*, float*, int64_t, int, \
                                        int64_t, int64_t, int64_t, int, float, float, hipStream_t);
MLA_INST(bf16_t, bf16_t)
MLA_INST(f16_t, f16_t)
MLA_INST(bf16_t, f8e5m2_t)
MLA_INST(bf16_t, f8e4m3_t)
MLA_INST(f16_t, f8e5m2_t)
MLA_INST(f16_t, f8e4m3_t)
#undef MLA_INST

}  // namespace semipd
