// GQA / MQA paged decode attention, stage 1, on the matrix cores (gfx950).
//
// One WAVE owns one (request, kv head, 16-q-head tile, kv split) and walks its token range in
// tiles of 32 tokens; four independent waves share a workgroup only to fill the CU.  Everything is
// transposed so that the softmax needs almost no cross-lane traffic (same trick as the prefill
// kernel):
//     S^T[token, head] = K_tile x Q^T          v_mfma_f32_16x16x32 (A = K rows straight from HBM
//                                              in operand layout, B = Q^T kept in registers)
//     O^T[d, head]    += V^T_tile x P^T        (A = V^T through ds_read_b64_tr_b16 from a
//                                              row-major LDS tile, B = P^T = the S^T registers)
// A lane holds head (lane & 15) for tokens {4*(lane>>4)+r, 16+4*(lane>>4)+r}: the row maximum is an
// in-lane max of 8 values plus two lane exchanges, P^T is already in B-operand order, and the O
// rescale is one scalar per lane.  Per 32 tokens a wave issues 16 MFMAs and ~60 VALU ops instead of
// the ~1100 VALU ops of the shuffle kernel, so the kernel stays on the HBM roofline for G = 4..16.
// Tile i+1 (K fragments, V rows, indices) is in flight in registers while tile i is computed.
#include "common.h"

namespace semipd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

union FragD {
  uint4 u;
  uint2 h[2];
  uint16_t e[8];
  s16x4 s[2];
  bf16x8_t b;
  f16x8_t f;
};

template <typename T> struct Mfma16;
template <> struct Mfma16<bf16_t> {
  __device__ static inline f32x4 mma(const FragD& a, const FragD& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct Mfma16<f16_t> {
  __device__ static inline f32x4 mma(const FragD& a, const FragD& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
  }
};

// K fragments of one 32-token tile (A operands of S^T = K Q^T) as they travel from HBM: 16 bytes per
// lane for pool rows in the activation type, 8 bytes for fp8 rows (expanded right before the MFMA)
template <int D, typename Raw> struct KRegs {
  Raw k[2][D / 32];
};

template <typename T, int D, typename KV>
__global__ void __launch_bounds__(256, D <= 128 ? 2 : 1)   // D = 256 needs > 256 registers: one workgroup per SIMD set instead of 672 B of scratch
decode_mfma_kernel(T* __restrict__ out, const T* __restrict__ q, const KV* __restrict__ k_buf,
                   const KV* __restrict__ v_buf, const int32_t* __restrict__ kv_indptr,
                   const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
                   int num_q_heads, int num_kv_heads, int group, int tiles_per_kv, int64_t q_stride,
                   int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride, int num_kv_splits,
                   int64_t total_items, float sm_scale, float logit_cap) {
  constexpr int KS = D / 32, DT = D / 16, CPR = D / 8, NV = CPR / 2;
  using KVT = KVTraits<T, KV>;
  using Raw = typename KVT::Raw;
  constexpr int PAD = ((D / 2) % 32 == 16) ? 0 : 64;  // row stride == 16 or 48 dwords (mod 64):
  constexpr int RS = D * 2 + PAD;                     // conflict-free ds_read_b64_tr_b16
  __shared__ __attribute__((aligned(16))) uint8_t v_lds_all[4][32 * RS];

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
  uint8_t* v_lds = v_lds_all[wave];

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

  // Q^T fragments (B operand): lane = head c16, d = ks*32 + q4*8 .. +8
  FragD qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks].u = make_uint4(0, 0, 0, 0);
    if (head_ok)
      qf[ks].u = *reinterpret_cast<const uint4*>(q + (int64_t)b * q_stride + (int64_t)(hq0 + c16) * D +
                                                 ks * 32 + q4 * 8);
  }

  f32x4 o_acc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) o_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int32_t* idx_base = kv_indices + kv_start;
  const int n_tiles = (s_end - s_begin + 31) >> 5;
  const int64_t k_head_off = (int64_t)hk * D + q4 * 8;
  const int64_t v_head_off = (int64_t)hk * D;

  auto load_idx = [&](int ti) -> int32_t {  // lane l (< 32) owns the index of token l of tile ti
    const int tok = s_begin + ti * 32 + (lane & 31);
    return (ti < n_tiles && tok < s_end) ? idx_base[tok] : 0;
  };
  auto issue_k = [&](KRegs<D, Raw>& r, int32_t idx_reg) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t row = (int64_t)__shfl(idx_reg, t * 16 + c16, 64) * kbuf_stride + k_head_off;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        r.k[t][ks] = KVT::load8(k_buf + row + ks * 32);
    }
  };
  Raw vreg[NV];  // V rows of the NEXT tile, in flight while the current tile is computed
  auto issue_v = [&](int32_t idx_reg, int ti) {
    const int base_tok = s_begin + ti * 32;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int it = j * 64 + lane;
      const int tok = it / CPR, ch = it - tok * CPR;
      const int64_t row = (int64_t)__shfl(idx_reg, tok, 64) * vbuf_stride + v_head_off;
      vreg[j] = KVT::zero();
      if (base_tok + tok < s_end)  // rows past the end must be zero: 0 * garbage could be NaN
        vreg[j] = KVT::load8(v_buf + row + ch * 8);
    }
  };

  // compute tile ti from K registers `r` and the V registers; as soon as the V registers are in
  // LDS they are re-used for the loads of tile ti+1 (idx_v = its indices)
  auto compute = [&](KRegs<D, Raw>& r, int ti, int32_t idx_v) {
    // ---- V rows -> LDS (row-major, this wave's private tile) ----
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int it = j * 64 + lane;
      const int tok = it / CPR, ch = it - tok * CPR;
      *reinterpret_cast<uint4*>(v_lds + tok * RS + ch * 16) = KVT::expand(vreg[j]);
    }
    if (ti + 1 < n_tiles) issue_v(idx_v, ti + 1);
    // ---- S^T = K Q^T ----
    f32x4 s_acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      s_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        FragD kf;
        kf.u = KVT::expand(r.k[t][ks]);
        s_acc[t] = Mfma16<T>::mma(kf, qf[ks], s_acc[t]);
      }
    }
    // ---- online softmax: lane = head c16, tokens t*16 + q4*4 + r ----
    const int base_tok = s_begin + ti * 32;
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
      for (int t = 0; t < DT; ++t) o_acc[t] *= alpha;
      m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
    FragD pf;
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
    // ---- O^T += V^T P^T : A operand through the transposing LDS read ----
    // 16-lane group q4 reads the 4x16 block {tokens q4*4 .. +3 (then 16 + ...)} x {d = dt*16 .. +15};
    // lane i of the group supplies row (i >> 2), columns 4*(i & 3) .. +3 and receives column i.
    const uint8_t* vrow = v_lds + (q4 * 4 + (c16 >> 2)) * RS + (c16 & 3) * 8;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      FragD a;
      a.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + dt * 32));
      a.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + 16 * RS + dt * 32));
      o_acc[dt] = Mfma16<T>::mma(a, pf, o_acc[dt]);
    }
  };

  KRegs<D, Raw> ra, rb;
  int32_t idx_cur = load_idx(0);
  issue_k(ra, idx_cur);
  issue_v(idx_cur, 0);
  int32_t idx_next = load_idx(1);
  for (int ti = 0; ti < n_tiles; ti += 2) {
    if (ti + 1 < n_tiles) issue_k(rb, idx_next);
    idx_cur = idx_next;            // indices of tile ti+1 (its V rows are issued inside compute)
    idx_next = load_idx(ti + 2);
    compute(ra, ti, idx_cur);
    if (ti + 1 < n_tiles) {
      if (ti + 2 < n_tiles) issue_k(ra, idx_next);
      idx_cur = idx_next;
      idx_next = load_idx(ti + 3);
      compute(rb, ti + 1, idx_cur);
    }
  }

  // ---- epilogue ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
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
