// The token walk of the GQA / MQA paged decode kernels on the matrix cores (gfx950): ONE wave, one
// (request, kv head, 16-q-head tile), one range of the request's tokens, in tiles of 32.  Shared by
// decode_mfma_kernel (decode_attention_mfma.hip: one wave per kv split, partials merged by stage 2) and
// decode_rope_attn_kernel (decode_attention_fused.hip: the splits are the waves of one workgroup and merge in LDS).
//
// Everything is transposed so that the softmax needs almost no cross-lane traffic (same trick as the prefill kernel):
//     S^T[token, head] = K_tile x Q^T          v_mfma_f32_16x16x32 (A = K rows straight from HBM
//                                              in operand layout, B = Q^T kept in registers)
//     O^T[d, head]    += V^T_tile x P^T        (A = V^T through ds_read_b64_tr_b16 from a
//                                              row-major LDS tile, B = P^T = the S^T registers)
// A lane holds head (lane & 15) for tokens {4*(lane>>4)+r, 16+4*(lane>>4)+r}: the row maximum is an
// in-lane max of 8 values plus two lane exchanges, P^T is already in B-operand order, and the O
// rescale is one scalar per lane.  Per 32 tokens a wave issues 16 MFMAs and ~60 VALU ops instead of
// the ~1100 VALU ops of the shuffle kernel, so the kernel stays on the HBM roofline for G = 4..16.
// Tile i+1 (K fragments, V rows, indices) is in flight in registers while tile i is computed.
#pragma once
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

// decode_attention.hip: merge of `num_kv_splits` stage-1 partials per (request, head) (decode_stage2_kernel)
int launch_decode_stage2(void* out, const float* attn_logits, const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                         int head_dim_v, int64_t o_stride, int num_kv_splits, int dtype, hipStream_t st);

// bytes of the wave-private V tile in LDS and its row stride
template <int D> struct DecodeWalkLds {
  static constexpr int PAD = ((D / 2) % 32 == 16) ? 0 : 64;  // row stride == 16 or 48 dwords (mod 64):
  static constexpr int RS = D * 2 + PAD;                     // conflict-free ds_read_b64_tr_b16
  static constexpr int BYTES = 32 * RS;
};

// Walks tokens [s_begin, s_end) of the sequence whose pool rows are idx_base[0 .. ).  An empty range is allowed (nothing
// is accumulated, l_tot = 0).
//   load_q(qf): fills the Q^T fragments (B operand): lane = head c16, d = ks*32 + q4*8 .. +8
//   mid():      called once before the Q fragments are fetched, after the loads of the first `early_tiles` (0, 1 or 2:
//               K and V rows of tile 0, K rows of tile 1) were issued -- the fused kernel rotates q / k there and stores
//               the new token's rows; a tile that holds the new token must not be asked for before that
// Results: o_acc (unnormalised O^T, lane = head c16, d = dt*16 + q4*4 + r), the running maximum m_run (per lane = head)
// and l_tot, the sum of the weights over the whole range (reduced over the four 16-lane groups).
template <typename T, int D, typename KV, typename QLoad, typename Mid>
__device__ __forceinline__ void decode_mfma_walk(uint8_t* v_lds, const KV* __restrict__ k_buf, const KV* __restrict__ v_buf,
                                                 const int32_t* __restrict__ idx_base, int s_begin, int s_end, int hk,
                                                 int64_t kbuf_stride, int64_t vbuf_stride, float sm_scale, float logit_cap,
                                                 int early_tiles, QLoad&& load_q, Mid&& mid, f32x4 (&o_acc)[D / 16],
                                                 float& m_run, float& l_tot) {
  constexpr int KS = D / 32, DT = D / 16, CPR = D / 8, NV = CPR / 2;
  using KVT = KVTraits<T, KV>;
  using Raw = typename KVT::Raw;
  constexpr int RS = DecodeWalkLds<D>::RS;
  const int lane = threadIdx.x & 63;
  const int c16 = lane & 15, q4 = lane >> 4;

#pragma unroll
  for (int t = 0; t < DT; ++t) o_acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  m_run = -INFINITY;
  float l_run = 0.f;

  const int n_tiles = s_end > s_begin ? (s_end - s_begin + 31) >> 5 : 0;
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

  FragD qf[KS];

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
  int32_t idx_next = load_idx(1);
  const bool k1_early = early_tiles >= 2 && n_tiles > 1;
  if (early_tiles >= 1) {
    issue_k(ra, idx_cur);
    issue_v(idx_cur, 0);
    if (k1_early) issue_k(rb, idx_next);
  }
  mid();
  load_q(qf);
  if (early_tiles < 1) {
    issue_k(ra, idx_cur);
    issue_v(idx_cur, 0);
  }
  for (int ti = 0; ti < n_tiles; ti += 2) {
    if (ti + 1 < n_tiles && !(ti == 0 && k1_early)) issue_k(rb, idx_next);
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

  l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
}

}  // namespace semipd
