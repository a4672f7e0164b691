// GQA / MQA paged decode attention, stage 1, on the matrix cores (gfx950).
//
// One WAVE owns one (request, kv head, 16-q-head tile, kv split) and walks its token range in
// tiles of 32 tokens (decode_mfma_walk.h: the walk, its layouts and why it is transposed); four
// independent waves share a workgroup only to fill the CU.
#include "decode_mfma_walk.h"

namespace semipd {

template <typename T, int D, typename KV>
__global__ void __launch_bounds__(256, D <= 128 ? 2 : 1)   // D = 256 needs > 256 registers: one workgroup per SIMD set instead of 672 B of scratch
decode_mfma_kernel(T* __restrict__ out, const T* __restrict__ q, const KV* __restrict__ k_buf,
                   const KV* __restrict__ v_buf, const int32_t* __restrict__ kv_indptr,
                   const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
                   int num_q_heads, int num_kv_heads, int group, int tiles_per_kv, int64_t q_stride,
                   int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride, int num_kv_splits,
                   int64_t total_items, float sm_scale, float logit_cap) {
  constexpr int KS = D / 32, DT = D / 16;
  __shared__ __attribute__((aligned(16))) uint8_t v_lds_all[4][DecodeWalkLds<D>::BYTES];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= total_items) return;  // no workgroup barrier below: waves are independent
  const int split = (int)(item % num_kv_splits);
  int64_t tmp = item / num_kv_splits;
  const int tile = (int)(tmp % tiles_per_kv);
  tmp /= tiles_per_kv;
  const int hk = (int)(tmp % num_kv_heads);
  const int b = (int)(tmp / num_kv_heads);
  const int h0 = tile * 16;
  const int heads = min(16, group - h0);
  const int hq0 = hk * group + h0;

  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  const bool head_ok = c16 < heads;
  if (s_end <= s_begin) {
    if (num_kv_splits == 1 && head_ok) {
      for (int d = q4; d < D; d += 4)
        out[(int64_t)b * o_stride + (int64_t)(hq0 + c16) * D + d] = Elem<T>::from_f(0.f);
    }
    return;
  }

  f32x4 o_acc[DT];
  float m_run, l_tot;
  decode_mfma_walk<T, D, KV>(
      v_lds_all[wave], k_buf, v_buf, kv_indices + kv_start, s_begin, s_end, hk, kbuf_stride, vbuf_stride, sm_scale,
      logit_cap, 2,
      [&](FragD (&qf)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          qf[ks].u = make_uint4(0, 0, 0, 0);
          if (head_ok)
            qf[ks].u = *reinterpret_cast<const uint4*>(q + (int64_t)b * q_stride + (int64_t)(hq0 + c16) * D +
                                                       ks * 32 + q4 * 8);
        }
      },
      [] {}, o_acc, m_run, l_tot);

  // ---- epilogue ----
  if (!head_ok) return;
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  const int hq = hq0 + c16;
  if (num_kv_splits == 1) {
    T* orow = out + (int64_t)b * o_stride + (int64_t)hq * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint2 w;
      w.x = (uint32_t)Elem<T>::from_f(o_acc[dt][0] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][1] * inv).v << 16);
      w.y = (uint32_t)Elem<T>::from_f(o_acc[dt][2] * inv).v | ((uint32_t)Elem<T>::from_f(o_acc[dt][3] * inv).v << 16);
      *reinterpret_cast<uint2*>(orow + dt * 16 + q4 * 4) = w;
    }
  } else {
    float* dst = attn_logits + (((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (D + 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dst[dt * 16 + q4 * 4 + rr] = o_acc[dt][rr] * inv;
    }
    if (q4 == 0) dst[D] = m_run + __logf(l_tot);
  }
}

template <typename T, typename KV>
int launch_decode_mfma(T* out, const T* q, const KV* k_buf, const KV* v_buf, const int32_t* kv_indptr,
                       const int32_t* kv_indices, float* attn_logits, int64_t batch, int Hq, int Hkv, int D,
                       int64_t q_stride, int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                       int splits, float sm_scale, float logit_cap, hipStream_t st) {
  const int group = Hq / Hkv;
  const int tiles = (group + 15) / 16;
  const int64_t total = batch * Hkv * tiles * splits;
  dim3 grid((unsigned)((total + 3) / 4)), block(256);
#define DM(DD)                                                                                      \
  hipLaunchKernelGGL((decode_mfma_kernel<T, DD, KV>), grid, block, 0, st, out, q, k_buf, v_buf, kv_indptr, \
                     kv_indices, attn_logits, Hq, Hkv, group, tiles, q_stride, o_stride, kbuf_stride,  \
                     vbuf_stride, splits, total, sm_scale, logit_cap)
  switch (D) {
    case 64: DM(64); break;
    case 96: DM(96); break;
    case 128: DM(128); break;
    case 256: DM(256); break;
    default: set_error("decode_mfma: head dim %d not instantiated", D); return SEMIPD_ESHAPE;
  }
#undef DM
  return launch_status("decode_mfma");
}

#define DM_INST(T, KV)                                                                                  \
  template int launch_decode_mfma<T, KV>(T*, const T*, const KV*, const KV*, const int32_t*, const int32_t*, \
                                         float*, int64_t, int, int, int, int64_t, int64_t, int64_t, int64_t,   \
                                         int, float, float, hipStream_t);
DM_INST(bf16_t, bf16_t)
DM_INST(f16_t, f16_t)
DM_INST(bf16_t, f8e5m2_t)
DM_INST(bf16_t, f8e4m3_t)
DM_INST(f16_t, f8e5m2_t)
DM_INST(f16_t, f8e4m3_t)
#undef DM_INST

}  // namespace semipd
