// Batched prefill ("extend") attention for gfx950: flash-attention forward over a
// paged prefix (gathered through kv_indices) followed by the causal triangle of the new tokens.
//
// Workgroup = 4 waves = 128 query rows of one (sequence, q head); each wave owns 32 rows.
// Everything is computed transposed so that the softmax statistics are lane-local:
//   S^T[kv,q] = K_tile (A operand, from LDS) x Q^T (B operand, registers)   v_mfma_f32_32x32x16
//   O^T[dv,q] += V^T_tile (A operand, from LDS) x P^T (B operand = the S^T registers, in place)
// A lane holds column q = lane&31 of S^T and O^T, so max/sum over kv are in-lane plus one
// exchange with lane^32, the O rescale factor is one scalar per lane, and P^T needs no
// cross-lane movement at all to become the B operand of the second MFMA (the kv-slot
// permutation of the C layout is applied to the V^T fragment addresses instead).
// K tiles are staged row-major (+16 B row pad: conflict-free ds_read_b128); V tiles are staged
// row-major too and transposed by the LDS itself on the way out (ds_read_b64_tr_b16).  The next
// KV tile is prefetched into registers while the current one is multiplied.  Masking code only
// runs on boundary tiles; the vector / logit-cap variants are separate instantiations so the hot
// loop carries no dead branches.
//
// Mirrors extend_attention_fwd (layers/attention/triton_ops/extend_attention.py:291-410).
#include "common.h"
#include "mfma_frag.h"

#include <algorithm>
#include <cstdlib>

namespace semipd {

// 8 consecutive elements of a row.  VEC: one 16-byte load (the chunk is either whole or absent);
// otherwise element-wise with zero fill beyond `valid`.
template <typename T, bool VEC>
__device__ inline Frag16 load_row8(const T* p, int valid) {
  Frag16 r;
  if (VEC) {
    r.u = valid > 0 ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) r.e[j] = j < valid ? p[j].v : (uint16_t)0;
  }
  return r;
}

// Two workgroups per CU (2 waves / SIMD, <= 256 VGPRs) wherever that fits without spilling: the second
// wave hides the LDS / softmax latency of the first.
typedef int ext_i32x4 __attribute__((ext_vector_type(4)));

// FAST (round 2; the PMC breakdown in profiles/r02_pmc_extend_attention.txt showed 13-16 VALU instructions per MFMA,
// most of them staging: per-load bounds branches, zero fills, 64-bit address arithmetic in two code paths): head
// dims equal to their padded sizes, rows in the activation type, no logit cap.  The NEW tokens' K / V rows are then
// fetched with buffer loads -- descriptor per (sequence, kv head) whose size ends at the last row the tile may
// see, so rows past the end read as zero in hardware; per-thread offsets are computed once, the tile offset is a
// scalar -- and the softmax scale is folded into the exponent's fma.
// custom_mask of extend_attention_fwd (extend_attention.py:291-307): one byte per (new token, kv position) of every
// sequence, rows of prefix + extend entries starting at mask_indptr[seq] (:91-92, :165-175, :234-245).  The prefix
// columns are read only when skip_prefix == 0 (SKIP_PREFIX_CUSTOM_MASK, :164); in the triangle the mask is AND-ed with
// j <= i: the reference visits the keys up to the end of the query's BLOCK_M tile there, so a bit above the diagonal
// would count or not with the tile size of the launch -- tree masks of speculative verification are sub-causal.
struct ExtMask {
  const uint8_t* mask;
  const int64_t* indptr;
  int skip_prefix;
};

template <typename T, int DKP, int DVP, bool VEC, bool CAP, typename KV = T, bool FAST = false, bool MASK = false>
__global__ void __launch_bounds__(256, (VEC && !CAP && DKP <= 128) ? 2 : 1)
extend_attn_kernel(T* __restrict__ out, const T* __restrict__ q_ext, const T* __restrict__ k_ext,
                   const T* __restrict__ v_ext, const KV* __restrict__ k_buf,
                   const KV* __restrict__ v_buf, const int32_t* __restrict__ qo_indptr,
                   const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices,
                   int group, int Dk, int Dv, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                   int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride, float sm_scale,
                   float logit_cap, ExtMask mk = ExtMask()) {
  static_assert(!(MASK && FAST), "the masked form is an instantiation of the general kernel");
  constexpr int BM = 128, BN = 64;
  constexpr int KS = DKP + 8;  // K tile row stride in elements (16 B pad)
  // V tile row stride in elements: row bytes == 64 (mod 128), i.e. 16 or 48 dwords (mod 64), which
  // makes the 4-row ds_read_b64_tr_b16 blocks conflict-free
  constexpr int VS = DVP + (((DVP / 2) % 32 == 16) ? 0 : 32);
  constexpr int KSTEPS = DKP / 16;
  constexpr int DVT = DVP / 32;
  __shared__ __attribute__((aligned(16))) uint16_t k_lds[BN * KS];
  __shared__ __attribute__((aligned(16))) uint16_t v_lds[BN * VS];

  const int seq = blockIdx.z, hq = blockIdx.x, q0 = (int)(gridDim.y - 1 - blockIdx.y) * BM;
  const int hk = hq / group;
  const int q_start = qo_indptr[seq];
  const int ext_len = qo_indptr[seq + 1] - q_start;
  if (q0 >= ext_len) return;
  const int kv_start = kv_indptr[seq];
  const int pre_len = kv_indptr[seq + 1] - kv_start;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const int q_local = q0 + wave * 32 + col;  // this lane's query row inside the extend part
  const bool q_valid = q_local < ext_len;
  const uint8_t* mk_row = nullptr;  // MASK: this lane's query row of the sequence's mask
  if constexpr (MASK) mk_row = mk.mask + mk.indptr[seq] + (int64_t)(q_valid ? q_local : 0) * (pre_len + ext_len);

  // Q^T fragments (B operand): lane holds Q[q_local][ks*16 + hi*8 .. +8]
  Frag16 qf[KSTEPS];
  {
    const T* qrow = q_ext + (int64_t)(q_start + q_local) * q_stride + (int64_t)hq * Dk;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      qf[ks] = load_row8<T, VEC>(qrow + d0, q_valid ? min(8, Dk - d0) : 0);
    }
  }

  f32x16 o_acc[DVT];
#pragma unroll
  for (int t = 0; t < DVT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running max in the log2 domain
  const float LOG2E = 1.4426950408889634f;
  const float qk_scale = sm_scale * LOG2E;

  // KV tiles: first the paged prefix (no mask), then the extend part (causal).  Tile i+1 is
  // fetched into registers while tile i is computed from LDS (issue early / write late).
  const int ext_end = min(ext_len, q0 + BM);
  const int n_pre_tiles = (pre_len + BN - 1) / BN;
  const int n_ext_tiles = (ext_end + BN - 1) / BN;
  const int n_tiles = n_pre_tiles + n_ext_tiles;
  constexpr int CHK = DKP / 8, CHV = DVP / 8;
  constexpr int NKI = (BN * CHK + 255) / 256, NVI = (BN * CHV + 255) / 256;
  Frag16 kreg[NKI], vreg[NVI];
  // paged prefix rows: activation type, or fp8 (8 bytes per 8 elements in flight, expanded when the
  // tile is written to LDS so that the loads stay in flight during the MFMAs)
  constexpr bool F8 = KVTraits<T, KV>::kF8;
  const KV* k_head = k_buf + (int64_t)hk * Dk;
  const KV* v_head = v_buf + (int64_t)hk * Dv;
  const T* ke_head = k_ext + (int64_t)q_start * k_stride + (int64_t)hk * Dk;
  const T* ve_head = v_ext + (int64_t)q_start * v_stride + (int64_t)hk * Dv;
  const int32_t* idx_base = kv_indices + kv_start;
  // FAST: the extend rows through buffer descriptors that end after row ext_end - 1 of this kv head
  __amdgpu_buffer_rsrc_t rsrc_k, rsrc_v;
  int kvo[NKI], vvo[NVI];
  if constexpr (FAST) {
    rsrc_k = __builtin_amdgcn_make_buffer_rsrc((void*)ke_head, 0, (int)(((int64_t)(ext_end - 1) * k_stride + Dk) * 2), 0x00020000);
    rsrc_v = __builtin_amdgcn_make_buffer_rsrc((void*)ve_head, 0, (int)(((int64_t)(ext_end - 1) * v_stride + Dv) * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int item = tid + i * 256;
      kvo[i] = (int)(((int64_t)(item / CHK) * k_stride + (item % CHK) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const int item = tid + i * 256;
      vvo[i] = (int)(((int64_t)(item / CHV) * v_stride + (item % CHV) * 8) * 2);
    }
  }

  // Per-thread staging slots: item = tid + i*256 -> (row r, 16-byte chunk c) of the tile.
  // The pool-slot indices of a prefix tile are loaded ONE ITERATION before its rows so that the
  // dependent index -> row address chain never stalls the wave in front of the MFMAs.
  int32_t idxk[NKI], idxv[NVI];
  auto load_idx = [&](int it) __attribute__((always_inline)) {
    if (it < n_pre_tiles) {
      const int n0 = it * BN;
#pragma unroll
      for (int i = 0; i < NKI; ++i) {
        const int n = n0 + (tid + i * 256) / CHK;
        idxk[i] = n < pre_len ? idx_base[n] : 0;
      }
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int n = n0 + (tid + i * 256) / CHV;
        idxv[i] = n < pre_len ? idx_base[n] : 0;
      }
    }
  };
  auto fetch = [&](int it) __attribute__((always_inline)) {
    if (it < n_pre_tiles) {  // paged prefix: rows through the indices loaded one iteration ago
      const int n0 = it * BN;
#pragma unroll
      for (int i = 0; i < NKI; ++i) {
        const int item = tid + i * 256;
        const int r = item / CHK, c = item - r * CHK;
        kreg[i].u = make_uint4(0, 0, 0, 0);
        if ((BN * CHK % 256 == 0 || item < BN * CHK) && n0 + r < pre_len) {
          if constexpr (F8) {
            const uint2 w = *reinterpret_cast<const uint2*>(k_head + (int64_t)idxk[i] * kbuf_stride + c * 8);
            kreg[i].u = make_uint4(w.x, w.y, 0u, 0u);
          } else {
            kreg[i] = load_row8<T, VEC>(reinterpret_cast<const T*>(k_head) + (int64_t)idxk[i] * kbuf_stride + c * 8,
                                        min(8, Dk - c * 8));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int item = tid + i * 256;
        const int r = item / CHV, c = item - r * CHV;
        vreg[i].u = make_uint4(0, 0, 0, 0);  // rows past the end stay zero: 0 * garbage could be NaN
        if ((BN * CHV % 256 == 0 || item < BN * CHV) && n0 + r < pre_len) {
          if constexpr (F8) {
            const uint2 w = *reinterpret_cast<const uint2*>(v_head + (int64_t)idxv[i] * vbuf_stride + c * 8);
            vreg[i].u = make_uint4(w.x, w.y, 0u, 0u);
          } else {
            vreg[i] = load_row8<T, VEC>(reinterpret_cast<const T*>(v_head) + (int64_t)idxv[i] * vbuf_stride + c * 8,
                                        min(8, Dv - c * 8));
          }
        }
      }
    } else if constexpr (FAST) {  // the new tokens: one buffer load per chunk, no VALU
      static_assert(!FAST || ((BN * CHK) % 256 == 0 && (BN * CHV) % 256 == 0), "whole passes of the 256 threads");
      const int n0 = (it - n_pre_tiles) * BN;
      const int ks_off = (int)((int64_t)n0 * k_stride * 2), vs_off = (int)((int64_t)n0 * v_stride * 2);
#pragma unroll
      for (int i = 0; i < NKI; ++i) {
        const ext_i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, kvo[i], ks_off, 0);
        kreg[i].u = make_uint4((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
      }
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const ext_i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, vvo[i], vs_off, 0);
        vreg[i].u = make_uint4((uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
      }
    } else {  // the new tokens: contiguous rows
      const int n0 = (it - n_pre_tiles) * BN;
#pragma unroll
      for (int i = 0; i < NKI; ++i) {
        const int item = tid + i * 256;
        const int r = item / CHK, c = item - r * CHK;
        kreg[i].u = make_uint4(0, 0, 0, 0);
        if ((BN * CHK % 256 == 0 || item < BN * CHK) && n0 + r < ext_end)
          kreg[i] = load_row8<T, VEC>(ke_head + (int64_t)(n0 + r) * k_stride + c * 8, min(8, Dk - c * 8));
      }
#pragma unroll
      for (int i = 0; i < NVI; ++i) {
        const int item = tid + i * 256;
        const int r = item / CHV, c = item - r * CHV;
        vreg[i].u = make_uint4(0, 0, 0, 0);
        if ((BN * CHV % 256 == 0 || item < BN * CHV) && n0 + r < ext_end)
          vreg[i] = load_row8<T, VEC>(ve_head + (int64_t)(n0 + r) * v_stride + c * 8, min(8, Dv - c * 8));
      }
    }
  };

  // 16-lane group g = lane >> 4 of the transposing read: dv block (g & 1) * 16 of a 32-wide tile,
  // kv rows 4*hi + ...; lane i of the group supplies row (i >> 2), columns 4*(i & 3) .. +3 and
  // receives column i.
  const int gi = lane & 15;
  const uint16_t* vbase = &v_lds[(hi * 4 + (gi >> 2)) * VS + ((lane >> 4) & 1) * 16 + (gi & 3) * 4];
  const uint16_t* kbase = &k_lds[col * KS + hi * 8];

  load_idx(0);
  if (n_tiles > 0) fetch(0);
  load_idx(1);
  for (int it = 0; it < n_tiles; ++it) {
    const bool pre = it < n_pre_tiles;
    const int n0 = (pre ? it : it - n_pre_tiles) * BN;
    const int n_end = pre ? pre_len : ext_end;
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int item = tid + i * 256;
      const int r = item / CHK, c = item - r * CHK;
      if (BN * CHK % 256 == 0 || item < BN * CHK) {
        uint4 u = kreg[i].u;
        if constexpr (F8) {
          if (pre) u = from_f8x8<KV, T>(make_uint2(u.x, u.y));
        }
        *reinterpret_cast<uint4*>(&k_lds[r * KS + c * 8]) = u;
      }
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const int item = tid + i * 256;
      const int r = item / CHV, c = item - r * CHV;
      if (BN * CHV % 256 == 0 || item < BN * CHV) {
        uint4 u = vreg[i].u;
        if constexpr (F8) {
          if (pre) u = from_f8x8<KV, T>(make_uint2(u.x, u.y));
        }
        *reinterpret_cast<uint4*>(&v_lds[r * VS + c * 8]) = u;
      }
    }
    __syncthreads();
    if (it + 1 < n_tiles) fetch(it + 1);  // in flight during the MFMAs below
    load_idx(it + 2);

    // this wave's rows are q0 + wave*32 .. +31: a causal tile entirely above them is all masked
    const int w_row0 = q0 + wave * 32;
    if (!pre && n0 > w_row0 + 31) continue;

    // ---- S^T = K Q^T : two 32-row kv tiles ----
    f32x16 s_acc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kt][r] = 0.f;
    // the two 32-row kv tiles alternate k-step by k-step: consecutive MFMAs never share an accumulator, so the
    // 8-pass latency of one hides behind the issue of the other (PMC: 24 % -> 34 % of the wave cycles were issue
    // stalls with the chains back to back)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        Frag16 a;
        a.u = *reinterpret_cast<const uint4*>(kbase + kt * 32 * KS + ks * 16);
        s_acc[kt] = Mfma<T>::mma(as_frag<T>(a), as_frag<T>(qf[ks]), s_acc[kt]);
      }
    }
    // ---- scale (+cap) and mask; masking only on boundary tiles (wave-uniform test) ----
    const bool need_mask = (n0 + BN > n_end) || (!pre && n0 + BN - 1 > w_row0);
    if (CAP && logit_cap > 0.f) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s_acc[kt][r] = logit_cap * tanhf(s_acc[kt][r] * sm_scale / logit_cap) * LOG2E;
    } else if constexpr (!FAST) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kt][r] *= qk_scale;
    }  // FAST: raw scores; qk_scale > 0 is applied to the row maximum and inside the exponent's fma below
    if constexpr (MASK) {
      const bool use_cm = !pre || !mk.skip_prefix;   // wave-uniform
      if (need_mask || use_cm) {
        const uint8_t* mrow = mk_row + (pre ? 0 : pre_len);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = n0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            bool ok = n < n_end && (pre || n <= q_local);
            if (use_cm && ok && q_valid) ok = mrow[n] != 0;
            s_acc[kt][r] = ok ? s_acc[kt][r] : -INFINITY;
          }
      }
    } else if (need_mask) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool ok = n < n_end && (pre || n <= q_local);
          s_acc[kt][r] = ok ? s_acc[kt][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if constexpr (FAST) mx *= qk_scale;   // -inf stays -inf
    // ---- online softmax (base 2) ----
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new > m_run)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
      m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;  // everything masked so far
    float psum = 0.f;
    Frag16 pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        // exp2(-inf) = 0
        const float p0 = __builtin_amdgcn_exp2f(FAST ? fmaf(s_acc[kt][r], qk_scale, -m_use) : s_acc[kt][r] - m_use);
        const float p1 = __builtin_amdgcn_exp2f(FAST ? fmaf(s_acc[kt][r + 1], qk_scale, -m_use) : s_acc[kt][r + 1] - m_use);
        psum += p0 + p1;
        pf[kt][r >> 3].w[(r & 7) >> 1] = pack2<T>(p0, p1);
      }
    }
    l_run += psum;
    // ---- O^T += V^T P^T : V^T fragments through the transposing LDS read ----
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int t = 0; t < DVT; ++t) {   // DVT independent accumulators in a row
          Frag16 a;
          const uint16_t* vp = vbase + (kt * 32 + s2 * 16) * VS + t * 32;
          a.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp));
          a.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 8 * VS));
          o_acc[t] = Mfma<T>::mma(as_frag<T>(a), as_frag<T>(pf[kt][s2]), o_acc[t]);
        }
      }
    }
  }

  // ---- epilogue: O[q][dv] = O^T / l ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  // a query whose every key a custom mask removed: NaN, as the reference's kernel writes (exp(-inf - -inf) in its rescale)
  // and as the one-wave kernel below does (acc / 0); without a mask every query sees at least itself
  const float inv = l_tot > 0.f ? 1.f / l_tot : (MASK ? __builtin_nanf("") : 0.f);
  if (q_valid) {
    T* orow = out + (int64_t)(q_start + q_local) * o_stride + (int64_t)hq * Dv;
#pragma unroll
    for (int t = 0; t < DVT; ++t) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dv0 = t * 32 + 8 * r4 + 4 * hi;
        if (VEC) {
          if (dv0 < Dv) {
            uint2 w;
            w.x = pack2<T>(o_acc[t][r4 * 4 + 0] * inv, o_acc[t][r4 * 4 + 1] * inv);
            w.y = pack2<T>(o_acc[t][r4 * 4 + 2] * inv, o_acc[t][r4 * 4 + 3] * inv);
            *reinterpret_cast<uint2*>(orow + dv0) = w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (dv0 + j < Dv) orow[dv0 + j] = Elem<T>::from_f(o_acc[t][r4 * 4 + j] * inv);
        }
      }
    }
  }
}

// Generic fallback (head dims without an MFMA instantiation, e.g. MLA 576/512 with prefix):
// one wave per (query token, q head).
template <typename T, typename KV>
__device__ __forceinline__ float pool_elem_to_f(const KV* p) {
  if constexpr (KVTraits<T, KV>::kF8) return F8Cvt<KV>::template unpack2<false>((uint32_t)p->v)[0];
  else return Elem<T>::to_f(*p);
}

template <typename T, typename KV = T>
__global__ void __launch_bounds__(64)
extend_attn_generic_kernel(T* __restrict__ out, const T* __restrict__ q_ext,
                           const T* __restrict__ k_ext, const T* __restrict__ v_ext,
                           const KV* __restrict__ k_buf, const KV* __restrict__ v_buf,
                           const int32_t* __restrict__ qo_indptr,
                           const int32_t* __restrict__ kv_indptr,
                           const int32_t* __restrict__ kv_indices, int group, int Dk, int Dv,
                           int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                           int64_t kbuf_stride, int64_t vbuf_stride, float sm_scale,
                           float logit_cap, ExtMask mk = ExtMask()) {
  constexpr int MAXR = 9;
  const int seq = blockIdx.z, hq = blockIdx.y, qi = blockIdx.x;
  const int hk = hq / group;
  const int q_start = qo_indptr[seq];
  const int ext_len = qo_indptr[seq + 1] - q_start;
  if (qi >= ext_len) return;
  const int kv_start = kv_indptr[seq];
  const int pre_len = kv_indptr[seq + 1] - kv_start;
  const int lane = threadIdx.x;
  float qf[MAXR], acc[MAXR];
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int d = lane + r * 64;
    qf[r] = d < Dk ? Elem<T>::to_f(q_ext[(int64_t)(q_start + qi) * q_stride + (int64_t)hq * Dk + d]) * sm_scale : 0.f;
    acc[r] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const int total = pre_len + qi + 1;
  const uint8_t* mk_row = mk.mask ? mk.mask + mk.indptr[seq] + (int64_t)qi * (pre_len + ext_len) : nullptr;
  for (int t = 0; t < total; ++t) {
    const bool pooled = t < pre_len;  // prefix rows live in the pool (possibly fp8), the rest in the extend tensors
    if (mk_row && !(pooled && mk.skip_prefix) && mk_row[t] == 0) continue;   // custom mask: the key does not exist for this row
    const T *kr = nullptr, *vr = nullptr;
    const KV *kp = nullptr, *vp = nullptr;
    if (pooled) {
      const int64_t idx = kv_indices[kv_start + t];
      kp = k_buf + idx * kbuf_stride + (int64_t)hk * Dk;
      vp = v_buf + idx * vbuf_stride + (int64_t)hk * Dv;
    } else {
      const int64_t row = q_start + (t - pre_len);
      kr = k_ext + row * k_stride + (int64_t)hk * Dk;
      vr = v_ext + row * v_stride + (int64_t)hk * Dv;
    }
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int dd = lane + r * 64;
      if (dd < Dk) d = fmaf(qf[r], pooled ? pool_elem_to_f<T, KV>(kp + dd) : Elem<T>::to_f(kr[dd]), d);
    }
    d = wave_sum(d);
    if (logit_cap > 0.f) d = logit_cap * tanhf(d / logit_cap);
    const float mn = fmaxf(m, d);
    const float sc = (m == -INFINITY) ? 0.f : __expf(m - mn);
    const float p = __expf(d - mn);
    l = l * sc + p;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int dd = lane + r * 64;
      const float vv = dd < Dv ? (pooled ? pool_elem_to_f<T, KV>(vp + dd) : Elem<T>::to_f(vr[dd])) : 0.f;
      acc[r] = acc[r] * sc + p * vv;
    }
    m = mn;
  }
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int dd = lane + r * 64;
    if (dd < Dv) out[(int64_t)(q_start + qi) * o_stride + (int64_t)hq * Dv + dd] = Elem<T>::from_f(acc[r] / l);
  }
}

template <typename T, bool VEC, bool CAP, typename KV = T, bool MASK = false>
static int launch_extend_variant(void* out, const void* q, const void* k, const void* v, const void* k_buf,
                                 const void* v_buf, const int32_t* qo_indptr, const int32_t* kv_indptr,
                                 const int32_t* kv_indices, int64_t batch, int Hq, int group, int Dk, int Dv,
                                 int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                 int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend, float sm_scale,
                                 float logit_cap, hipStream_t st, ExtMask mk = ExtMask()) {
  const int dkp = (Dk + 15) / 16 * 16, dvp = (Dv + 31) / 32 * 32;
  // x = head (fastest), y = query tile: dispatch order is x-major, so every head of the longest
  // (last, causal) query tile starts first and the short tiles fill the tail
  dim3 grid((unsigned)Hq, (unsigned)((max_len_extend + 127) / 128), (unsigned)batch), block(256);
#define EXT_(DKP, DVP, FASTV)                                                                             \
  hipLaunchKernelGGL((extend_attn_kernel<T, DKP, DVP, VEC, CAP, KV, FASTV, MASK>), grid, block, 0, st, (T*)out, \
                     (const T*)q, (const T*)k, (const T*)v, (const KV*)k_buf, (const KV*)v_buf,            \
                     qo_indptr, kv_indptr, kv_indices, group, Dk, Dv, q_stride, k_stride, v_stride,        \
                     o_stride, kbuf_stride, vbuf_stride, sm_scale, logit_cap, mk)
#define EXT(DKP, DVP) EXT_(DKP, DVP, false)
  // FAST: exact head dims, activation-type rows, no cap, extend rows addressable by a 32-bit buffer offset
  constexpr bool kFastType = VEC && !CAP && !MASK && !KVTraits<T, KV>::kF8;
  const bool fast = kFastType && (int64_t)max_len_extend * std::max(k_stride, v_stride) * 2 < (1ll << 31);
  if (dkp <= 16 && dvp <= 32) EXT(16, 32);
  else if (dkp <= 64 && dvp <= 64) {
    if constexpr (kFastType) { if (fast && Dk == 64 && Dv == 64) EXT_(64, 64, true); else EXT(64, 64); }
    else EXT(64, 64);
  }
  else if (dkp <= 96 && dvp <= 96) EXT(96, 96);
  else if (dkp <= 128 && dvp <= 128) {
    if constexpr (kFastType) { if (fast && Dk == 128 && Dv == 128) EXT_(128, 128, true); else EXT(128, 128); }
    else EXT(128, 128);
  }
  else if (dkp <= 192 && dvp <= 128) {
    if constexpr (KVTraits<T, KV>::kF8) return 1;  // MLA prefill keeps its rows in the activation type
    else if constexpr (kFastType) { if (fast && Dk == 192 && Dv == 128) EXT_(192, 128, true); else EXT(192, 128); }
    else EXT(192, 128);
  } else return 1;  // no MFMA instantiation
#undef EXT_
#undef EXT
  return 0;
}

template <typename T>
int launch_extend_shared_kv(void* out, const void* q, const void* k, const void* v, const void* k_buf, const void* v_buf,
                            const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                            int Hq, int Hkv, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                            int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend, float sm_scale, hipStream_t st);

template <typename T>
static int run_extend(void* out, const void* q, const void* k, const void* v, const void* k_buf,
                      const void* v_buf, const int32_t* qo_indptr, const int32_t* kv_indptr,
                      const int32_t* kv_indices, int64_t batch, int Hq, int Hkv, int Dk, int Dv,
                      int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                      int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend, float sm_scale,
                      float logit_cap, int dtype, int kv_dtype, hipStream_t st, ExtMask mk = ExtMask()) {
  const int group = Hq / Hkv;
  const bool vec_ok = Dk % 8 == 0 && Dv % 8 == 0 && q_stride % 8 == 0 && k_stride % 8 == 0 &&
                      v_stride % 8 == 0 && o_stride % 4 == 0 && kbuf_stride % 8 == 0 &&
                      vbuf_stride % 8 == 0 && aligned16(q) && aligned16(k) && aligned16(v) &&
                      aligned16(k_buf) && aligned16(v_buf) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  if (kv_dtype != dtype) {
    // fp8 prefix rows (mem_cache/memory_pool.py:205-209): vectorised, cap-free instantiations only
    SEMIPD_CHECK_ARG(kv_dtype == SEMIPD_F8E5M2 || kv_dtype == SEMIPD_F8E4M3, SEMIPD_EDTYPE,
                     "extend_attention: unsupported kv_dtype %d", kv_dtype);
    if (mk.mask || !(vec_ok && !(logit_cap > 0.f) && Dk <= 128 && Dv <= 128)) {
      // everything else (MLA's 576 / 512 latent rows behind a chunked prefill, logit caps, ragged heads): the
      // one-wave-per-(token, head) kernel reads pool rows element by element in either storage type
      dim3 g2((unsigned)max_len_extend, (unsigned)Hq, (unsigned)batch);
      if (kv_dtype == SEMIPD_F8E5M2)
        hipLaunchKernelGGL((extend_attn_generic_kernel<T, f8e5m2_t>), g2, dim3(64), 0, st, (T*)out, (const T*)q,
                           (const T*)k, (const T*)v, (const f8e5m2_t*)k_buf, (const f8e5m2_t*)v_buf, qo_indptr, kv_indptr,
                           kv_indices, group, Dk, Dv, q_stride, k_stride, v_stride, o_stride, kbuf_stride, vbuf_stride,
                           sm_scale, logit_cap, mk);
      else
        hipLaunchKernelGGL((extend_attn_generic_kernel<T, f8e4m3_t>), g2, dim3(64), 0, st, (T*)out, (const T*)q,
                           (const T*)k, (const T*)v, (const f8e4m3_t*)k_buf, (const f8e4m3_t*)v_buf, qo_indptr, kv_indptr,
                           kv_indices, group, Dk, Dv, q_stride, k_stride, v_stride, o_stride, kbuf_stride, vbuf_stride,
                           sm_scale, logit_cap, mk);
      return launch_status("extend_attention(generic, fp8 pool)");
    }
    int miss8;
    if (kv_dtype == SEMIPD_F8E5M2)
      miss8 = launch_extend_variant<T, true, false, f8e5m2_t>(out, q, k, v, k_buf, v_buf, qo_indptr, kv_indptr,
                                                              kv_indices, batch, Hq, group, Dk, Dv, q_stride, k_stride,
                                                              v_stride, o_stride, kbuf_stride, vbuf_stride,
                                                              max_len_extend, sm_scale, logit_cap, st);
    else
      miss8 = launch_extend_variant<T, true, false, f8e4m3_t>(out, q, k, v, k_buf, v_buf, qo_indptr, kv_indptr,
                                                              kv_indices, batch, Hq, group, Dk, Dv, q_stride, k_stride,
                                                              v_stride, o_stride, kbuf_stride, vbuf_stride,
                                                              max_len_extend, sm_scale, logit_cap, st);
    SEMIPD_CHECK_ARG(!miss8, SEMIPD_ESHAPE, "extend_attention: no fp8 instantiation for these head sizes");
    return launch_status("extend_attention");
  }
  // Llama-shaped heads (128 / 128, rows in the activation type, no cap): the shared-KV kernel
  // (extend_attention_shared_kv.hip); SEMIPD_EXTEND_SHARED_KV=0 keeps the one-head-per-workgroup kernel
  static const bool shared_kv_on = [] { const char* e = getenv("SEMIPD_EXTEND_SHARED_KV"); return !(e && e[0] == '0'); }();
  if (shared_kv_on && !mk.mask && vec_ok && !(logit_cap > 0.f) && Dk == 128 && Dv == 128 &&
      (int64_t)max_len_extend * std::max(k_stride, v_stride) * 2 < (1ll << 31)) {
    if (launch_extend_shared_kv<T>(out, q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, Hq, Hkv, q_stride,
                                   k_stride, v_stride, o_stride, kbuf_stride, vbuf_stride, max_len_extend, sm_scale, st) == 0)
      return launch_status("extend_attention(shared kv)");
  }
  int miss;
#define VARIANT(V, C)                                                                                 \
  launch_extend_variant<T, V, C>(out, q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, Hq, \
                                 group, Dk, Dv, q_stride, k_stride, v_stride, o_stride, kbuf_stride,      \
                                 vbuf_stride, max_len_extend, sm_scale, logit_cap, st)
  // a custom mask: the general tile kernel's masked instantiation (vectorised rows; with or without a cap), else the
  // one-wave-per-(token, head) kernel, which reads the mask byte of every key it visits
  if (mk.mask)
    miss = vec_ok ? launch_extend_variant<T, true, true, T, true>(out, q, k, v, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices,
                                                                  batch, Hq, group, Dk, Dv, q_stride, k_stride, v_stride,
                                                                  o_stride, kbuf_stride, vbuf_stride, max_len_extend,
                                                                  sm_scale, logit_cap, st, mk)
                  : 1;
  else if (vec_ok && !(logit_cap > 0.f)) miss = VARIANT(true, false);
  else if (vec_ok) miss = VARIANT(true, true);
  else miss = VARIANT(false, true);
#undef VARIANT
  if (miss) {
    dim3 g2((unsigned)max_len_extend, (unsigned)Hq, (unsigned)batch);
    hipLaunchKernelGGL((extend_attn_generic_kernel<T>), g2, dim3(64), 0, st, (T*)out, (const T*)q,
                       (const T*)k, (const T*)v, (const T*)k_buf, (const T*)v_buf, qo_indptr,
                       kv_indptr, kv_indices, group, Dk, Dv, q_stride, k_stride, v_stride, o_stride,
                       kbuf_stride, vbuf_stride, sm_scale, logit_cap, mk);
  }
  return launch_status("extend_attention");
}

}  // namespace semipd

using namespace semipd;

extern "C" int semipd_extend_attention(void* out, const void* q_extend, const void* k_extend,
                                       const void* v_extend, const void* k_buf, const void* v_buf,
                                       const int32_t* qo_indptr, const int32_t* kv_indptr,
                                       const int32_t* kv_indices, int64_t batch, int num_q_heads,
                                       int num_kv_heads, int head_dim_k, int head_dim_v,
                                       int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                       int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                                       int max_len_extend, float sm_scale, float logit_cap, int dtype,
                                       int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && num_q_heads > 0 && num_kv_heads > 0 && head_dim_k > 0 &&
                       head_dim_v > 0 && max_len_extend >= 0,
                   SEMIPD_EINVAL, "extend_attention: bad sizes");
  SEMIPD_CHECK_ARG(num_q_heads % num_kv_heads == 0, SEMIPD_ESHAPE,
                   "extend_attention: Hq %d not a multiple of Hkv %d", num_q_heads, num_kv_heads);
  SEMIPD_CHECK_ARG(head_dim_k <= 576 && head_dim_v <= 576, SEMIPD_ESHAPE,
                   "extend_attention: head dims up to 576 supported");
  SEMIPD_CHECK_ARG(batch <= 65535 && num_q_heads <= 65535, SEMIPD_EINVAL,
                   "extend_attention: grid too large");
  if (batch == 0 || max_len_extend == 0) return 0;
  SEMIPD_CHECK_ARG(out && q_extend && k_extend && v_extend && qo_indptr && kv_indptr, SEMIPD_EINVAL,
                   "extend_attention: null pointer");
  SEMIPD_DISPATCH_HALF(dtype, T, return run_extend<T>(out, q_extend, k_extend, v_extend, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, num_q_heads, num_kv_heads, head_dim_k, head_dim_v, q_stride, k_stride, v_stride, o_stride, kbuf_stride, vbuf_stride, max_len_extend, sm_scale, logit_cap, dtype, kv_dtype, as_stream(stream)));
  return 0;
}

/* The same launch under extend_attention_fwd's custom_mask / mask_indptr / skip_prefix_custom_mask arguments
 * (extend_attention.py:291-307; the target-verify step of speculative decoding, triton_backend.py:136-149). */
extern "C" int semipd_extend_attention_masked(void* out, const void* q_extend, const void* k_extend,
                                              const void* v_extend, const void* k_buf, const void* v_buf,
                                              const int32_t* qo_indptr, const int32_t* kv_indptr,
                                              const int32_t* kv_indices, const uint8_t* custom_mask,
                                              const int64_t* mask_indptr, int skip_prefix_custom_mask, int64_t batch,
                                              int num_q_heads, int num_kv_heads, int head_dim_k, int head_dim_v,
                                              int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                                              int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend,
                                              float sm_scale, float logit_cap, int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && num_q_heads > 0 && num_kv_heads > 0 && head_dim_k > 0 && head_dim_v > 0 &&
                       max_len_extend >= 0,
                   SEMIPD_EINVAL, "extend_attention_masked: bad sizes");
  SEMIPD_CHECK_ARG(num_q_heads % num_kv_heads == 0, SEMIPD_ESHAPE,
                   "extend_attention_masked: Hq %d not a multiple of Hkv %d", num_q_heads, num_kv_heads);
  SEMIPD_CHECK_ARG(head_dim_k <= 576 && head_dim_v <= 576, SEMIPD_ESHAPE,
                   "extend_attention_masked: head dims up to 576 supported");
  SEMIPD_CHECK_ARG(batch <= 65535 && num_q_heads <= 65535, SEMIPD_EINVAL, "extend_attention_masked: grid too large");
  if (batch == 0 || max_len_extend == 0) return 0;
  SEMIPD_CHECK_ARG(out && q_extend && k_extend && v_extend && qo_indptr && kv_indptr, SEMIPD_EINVAL,
                   "extend_attention_masked: null pointer");
  SEMIPD_CHECK_ARG(custom_mask && mask_indptr, SEMIPD_EINVAL,
                   "extend_attention_masked: custom_mask and mask_indptr are both required (semipd_extend_attention "
                   "is the unmasked launch)");
  ExtMask mk;
  mk.mask = custom_mask;
  mk.indptr = mask_indptr;
  mk.skip_prefix = skip_prefix_custom_mask ? 1 : 0;
  SEMIPD_DISPATCH_HALF(dtype, T, return run_extend<T>(out, q_extend, k_extend, v_extend, k_buf, v_buf, qo_indptr, kv_indptr, kv_indices, batch, num_q_heads, num_kv_heads, head_dim_k, head_dim_v, q_stride, k_stride, v_stride, o_stride, kbuf_stride, vbuf_stride, max_len_extend, sm_scale, logit_cap, dtype, kv_dtype, as_stream(stream), mk));
  return 0;
}
