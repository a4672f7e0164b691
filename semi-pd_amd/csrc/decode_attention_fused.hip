// One launch between the qkv GEMM and o_proj of a decode step (GQA / MQA, Llama-shaped heads), gfx950:
//     qkv K-slice planes --sum, round--> q, k, v of the new token
//     RoPE (neox, whole head) on q and k;  k, v -> the KV pool row loc[b]
//     paged attention of the 4..16 q heads of one kv head over the request's tokens (the new one included)
//     merge of the kv splits
// replacing semipd_rope_kv_store_planes + decode_mfma_kernel + decode_stage2_kernel: three dependent launches of ~5, ~22
// and ~4.5 us per decoder layer at 32 requests, where the two small ones cost what a dependent graph node costs, not what
// they compute, and the split partials (8.7 MB per layer at 32 requests x 8 splits) made a round trip through HBM.
//
// A WORKGROUP owns one (request, kv head); its NW waves are the kv splits (same ranges as decode_mfma_kernel at
// num_kv_splits = NW, walked by the same code: decode_mfma_walk.h).  (Small batches: Z workgroups per pair, each merging
// its NW splits into ONE stage-1 partial; decode_stage2_kernel then merges Z partials per head instead of NW * Z.)
//   1. the lanes that rotate ask for their planes and cos / sin values, then every wave asks for its first tile of K / V
//      rows and the K rows of the second (unless they hold the new token);
//   2. the wave whose range ends with the new token sums the planes of k and v, rotates k and stores both pool rows
//      (and waits for its stores: it is the only reader of those rows in this launch); the other waves rotate q into LDS;
//   3. a raw barrier on the LDS writes only: no wave waits there for the K / V rows it has in flight;
//   4. the walk; each wave leaves its normalised partial and log-sum-exp in its own (now free) V tile in LDS;
//   5. barrier; the merge of stage 2 (weights exp(lse_s - max lse), sums in split order), one output per lane.
// Same arithmetic in the same order as the three launches: the output has their bits (tests/test_gpu_ops.py).
#include "decode_mfma_walk.h"

#include "../../include/semipd.h"

namespace semipd {

template <typename T, int D, typename KV, int NW>
__global__ void __launch_bounds__(NW * 64, 2)
decode_rope_attn_kernel(T* __restrict__ out, float* __restrict__ attn_logits, const float* __restrict__ planes, int n_planes,
                        int64_t plane_elems,
                        int64_t row_elems, KV* __restrict__ k_buf, KV* __restrict__ v_buf, const int64_t* __restrict__ loc,
                        const float* __restrict__ cache, const int64_t* __restrict__ positions,
                        const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices, int num_q_heads,
                        int num_kv_heads, int group, int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                        float sm_scale, float logit_cap) {
  constexpr int KS = D / 32, DT = D / 16, V = 8;
  constexpr int PART_STRIDE = D + 4;                       // floats per head of a wave's partial (16-byte aligned rows)
  constexpr int WAVE_LDS = DecodeWalkLds<D>::BYTES >= 16 * PART_STRIDE * 4 ? DecodeWalkLds<D>::BYTES : 16 * PART_STRIDE * 4;
  __shared__ __attribute__((aligned(16))) uint8_t wave_lds[NW][WAVE_LDS];   // V tile of the walk, then the partial
  __shared__ __attribute__((aligned(16))) uint16_t q_lds[16 * D];           // rotated q, [head][d]
  __shared__ float lse_lds[NW][16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.x / num_kv_heads, hk = blockIdx.x - b * num_kv_heads;
  const int hq0 = hk * group;
  const bool head_ok = c16 < group;

  // gridDim.y = Z workgroups per (request, kv head): NW * Z kv splits in all, workgroup z owns splits z * NW .. + NW - 1.
  // Z = 1: the merged result is the output.  Z > 1 (batches too small to fill the chip with one workgroup per pair): the
  // workgroup's merge is ONE partial of attn_logits[b, hq, z, :] for decode_stage2_kernel -- Z partials per head instead of
  // NW * Z, and still no separate RoPE launch.
  const int Z = gridDim.y, z = blockIdx.y;
  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + NW * Z - 1) / (NW * Z);
  const int s_begin = per_split * (z * NW + wave);
  const int s_end = min(s_begin + per_split, seq_len);
  // The new token is the last one of the sequence.  The wave whose range ends with it rotates k, copies v and stores both
  // rows itself (its later loads of that row are ordered behind its own stores by a vmcnt(0)); the other waves rotate q
  // into LDS.  Nobody else reads the new rows, so the workgroup meets on the LDS writes only (a raw barrier: no wave
  // waits for its K / V rows in flight).  Tiles that cannot reach the new token are asked for before all that.
  const int owner_g = seq_len > 0 ? (seq_len - 1) / per_split : NW * Z - 1;
  const int owner = owner_g / NW == z ? owner_g - z * NW : -1;     // -1: the new token belongs to another workgroup
  constexpr int half = D / 2, IPH = half / V;                 // items of 8 pairs per head
  // a wave that rotates asks for its planes first and for its tiles after the barrier: loads return in order, and behind
  // the K / V rows of two tiles the planes would arrive a memory latency later -- with every wave of every CU of the
  // launch waiting at the same barrier.  The other waves have two tiles in flight by then.
  const bool rotates = wave == owner || ((owner < 0 || wave < owner) ? wave : wave - 1) * 64 < group * IPH;
  const int early_tiles = rotates ? 0 : ((s_begin >= seq_len || s_begin + 64 < seq_len) ? 2 : (s_begin + 32 < seq_len ? 1 : 0));

  // ---- the rotation ----
  constexpr int ROLE_NONE = 0, ROLE_Q = 1, ROLE_K = 2, ROLE_V = 3, PRE = 4;   // PRE planes travel before the tiles
  int role = ROLE_NONE, item_h = 0, i0 = 0;
  if (wave == owner) {
    if (lane < IPH) role = ROLE_K, i0 = lane * V;
    else if (lane < IPH + D / V) role = ROLE_V, i0 = (lane - IPH) * V;
  } else {
    const int it = ((owner < 0 || wave < owner) ? wave : wave - 1) * 64 + lane;   // 16 heads x 8 items <= 2 of the >= 3 other waves
    if (it < group * IPH) role = ROLE_Q, item_h = it / IPH, i0 = (it - item_h * IPH) * V;
  }
  const float* pa = planes + (int64_t)b * row_elems + i0 +
                    (int64_t)(role == ROLE_Q ? hq0 + item_h : role == ROLE_K ? num_q_heads + hk : num_q_heads + num_kv_heads + hk) * D;
  const float* cs = cache + positions[b] * D;
  float4 pl[PRE][4], csr[4];
  auto rotation_loads = [&]() {
    if (role == ROLE_NONE) return;
#pragma unroll
    for (int z = 0; z < PRE; ++z) {
      if (z < n_planes) {
        const float* pz = pa + (int64_t)z * plane_elems;
        pl[z][0] = *reinterpret_cast<const float4*>(pz);
        pl[z][1] = *reinterpret_cast<const float4*>(pz + 4);
        if (role != ROLE_V) {
          pl[z][2] = *reinterpret_cast<const float4*>(pz + half);
          pl[z][3] = *reinterpret_cast<const float4*>(pz + half + 4);
        }
      }
    }
    if (role != ROLE_V) {
      csr[0] = *reinterpret_cast<const float4*>(cs + i0);
      csr[1] = *reinterpret_cast<const float4*>(cs + i0 + 4);
      csr[2] = *reinterpret_cast<const float4*>(cs + half + i0);
      csr[3] = *reinterpret_cast<const float4*>(cs + half + i0 + 4);
    }
  };
  auto rotate_and_store = [&]() {
    if (role != ROLE_NONE) {
      // planes summed in slice order (planes_sum_f4's order: the bits of splitk_planes_reduce)
      const int npos = role == ROLE_V ? 2 : 4;
      float4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < npos) {
          acc[i] = pl[0][i];
#pragma unroll
          for (int z = 1; z < PRE; ++z)
            if (z < n_planes) { acc[i].x += pl[z][i].x; acc[i].y += pl[z][i].y; acc[i].z += pl[z][i].z; acc[i].w += pl[z][i].w; }
          for (int z = PRE; z < n_planes; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(pa + (int64_t)z * plane_elems + (i & 1) * 4 + (i >> 1) * half);
            acc[i].x += t.x; acc[i].y += t.y; acc[i].z += t.z; acc[i].w += t.w;
          }
        }
      }
      const float fa[8] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w, acc[1].x, acc[1].y, acc[1].z, acc[1].w};
      if (role == ROLE_V) {
        Vec16<T> a;
#pragma unroll
        for (int j = 0; j < V; ++j) a.e[j] = Elem<T>::from_f(fa[j]);
        KVTraits<T, KV>::store8(v_buf + loc[b] * vbuf_stride + (int64_t)hk * D + i0, a);
      } else {
        const float fb[8] = {acc[2].x, acc[2].y, acc[2].z, acc[2].w, acc[3].x, acc[3].y, acc[3].z, acc[3].w};
        const float cc[8] = {csr[0].x, csr[0].y, csr[0].z, csr[0].w, csr[1].x, csr[1].y, csr[1].z, csr[1].w};
        const float sn[8] = {csr[2].x, csr[2].y, csr[2].z, csr[2].w, csr[3].x, csr[3].y, csr[3].z, csr[3].w};
        Vec16<T> oa, ob;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float x1 = Elem<T>::to_f(Elem<T>::from_f(fa[j])), x2 = Elem<T>::to_f(Elem<T>::from_f(fb[j]));
          float r1, r2;
          rope_pair(x1, x2, cc[j], sn[j], r1, r2);
          oa.e[j] = Elem<T>::from_f(r1);
          ob.e[j] = Elem<T>::from_f(r2);
        }
        if (role == ROLE_Q) {
          T* qh = reinterpret_cast<T*>(q_lds) + item_h * D;
          store16(qh + i0, oa);
          store16(qh + half + i0, ob);
        } else {
          KV* kh = k_buf + loc[b] * kbuf_stride + (int64_t)hk * D;
          KVTraits<T, KV>::store8(kh + i0, oa);
          KVTraits<T, KV>::store8(kh + half + i0, ob);
        }
      }
    }
    if (wave == owner) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows are in the L2 this wave loads through
  };

  f32x4 o_acc[DT];
  float m_run, l_tot;
  rotation_loads();
  decode_mfma_walk<T, D, KV>(
      wave_lds[wave], k_buf, v_buf, kv_indices + kv_start, s_begin, s_end, hk, kbuf_stride, vbuf_stride, sm_scale,
      logit_cap, early_tiles,
      [&](FragD (&qf)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          qf[ks].u = make_uint4(0, 0, 0, 0);
          if (head_ok) qf[ks].u = *reinterpret_cast<const uint4*>(q_lds + c16 * D + ks * 32 + q4 * 8);
        }
      },
      [&] {
        rotate_and_store();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's share of q is in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      },
      o_acc, m_run, l_tot);

  // ---- this wave's partial: what decode_mfma_kernel writes to attn_logits ----
  {
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    float* part = reinterpret_cast<float*>(wave_lds[wave]) + c16 * PART_STRIDE;
    if (head_ok) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        float4 w;
        w.x = o_acc[dt][0] * inv; w.y = o_acc[dt][1] * inv; w.z = o_acc[dt][2] * inv; w.w = o_acc[dt][3] * inv;
        *reinterpret_cast<float4*>(part + dt * 16 + q4 * 4) = w;
      }
      if (q4 == 0) lse_lds[wave][c16] = m_run + __logf(l_tot);
    }
  }
  __syncthreads();

  // ---- merge (decode_stage2_kernel): this workgroup's splits [0, n_valid) are non-empty ----
  const int first = z * NW * per_split;
  const int n_valid = (per_split > 0 && first < seq_len) ? min(NW, (seq_len - first + per_split - 1) / per_split) : 0;
  for (int idx = tid; idx < group * D; idx += NW * 64) {
    const int h = idx / D, d = idx - h * D;
    float e_max = -INFINITY;
    for (int s = 0; s < n_valid; ++s) e_max = fmaxf(e_max, lse_lds[s][h]);
    float w[NW];
    float e_sum = 0.f;
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      w[s] = s < n_valid ? __expf(lse_lds[s][h] - e_max) : 0.f;
      if (s < n_valid) e_sum += w[s];
    }
    const float inv = e_sum > 0.f ? 1.f / e_sum : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < NW; ++s)
      if (s < n_valid) acc = __builtin_fmaf(w[s], reinterpret_cast<const float*>(wave_lds[s])[h * PART_STRIDE + d], acc);
    if (Z == 1) {
      out[(int64_t)b * o_stride + (int64_t)(hq0 + h) * D + d] = Elem<T>::from_f(acc * inv);
    } else {
      // a partial in stage 1's form: normalised row + log-sum-exp; an empty workgroup (fewer tokens than splits) leaves
      // zeros and -inf, which stage 2 weighs with exp(-inf) = 0 wherever its own count of valid splits reaches it
      float* dst = attn_logits + (((int64_t)b * num_q_heads + hq0 + h) * Z + z) * (D + 1);
      dst[d] = acc * inv;
      if (d == 0) dst[D] = n_valid > 0 ? e_max + __logf(e_sum) : -INFINITY;
    }
  }
}

template <typename T, typename KV>
static int launch_decode_rope_attn(void* out, float* attn_logits, int zsplits, const float* planes, int n_planes, int64_t plane_elems, void* k_buf, void* v_buf,
                                   const int64_t* loc, const float* cache, const int64_t* positions, const int32_t* kv_indptr,
                                   const int32_t* kv_indices, int64_t batch, int Hq, int Hkv, int D, int64_t o_stride,
                                   int64_t kbuf_stride, int64_t vbuf_stride, int waves, float sm_scale, float logit_cap,
                                   hipStream_t st) {
  const int group = Hq / Hkv;
  const int64_t row_elems = (int64_t)(Hq + 2 * Hkv) * D;
  dim3 grid((unsigned)(batch * Hkv), (unsigned)zsplits), block((unsigned)(waves * 64));
#define DF(DD, NW)                                                                                                     \
  hipLaunchKernelGGL((decode_rope_attn_kernel<T, DD, KV, NW>), grid, block, 0, st, (T*)out, attn_logits, planes, n_planes, plane_elems, \
                     row_elems, (KV*)k_buf, (KV*)v_buf, loc, cache, positions, kv_indptr, kv_indices, Hq, Hkv, group,      \
                     o_stride, kbuf_stride, vbuf_stride, sm_scale, logit_cap)
  if (D == 128 && waves == 8) DF(128, 8);
  else if (D == 128 && waves == 4) DF(128, 4);
  else if (D == 64 && waves == 8) DF(64, 8);
  else if (D == 64 && waves == 4) DF(64, 4);
  else {
    set_error("decode_rope_attention_planes: head size %d with %d waves is not instantiated (64 / 128, 4 / 8)", D, waves);
    return SEMIPD_ESHAPE;
  }
#undef DF
  return launch_status("decode_rope_attention_planes");
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_decode_rope_attention_planes_supported(int num_q_heads, int num_kv_heads, int head_size, int dtype, int kv_dtype) {
  if (num_q_heads <= 0 || num_kv_heads <= 0 || num_q_heads % num_kv_heads) return 0;
  const int group = num_q_heads / num_kv_heads;
  if (group < 2 || group > 16 || (head_size != 64 && head_size != 128)) return 0;   // one q head per kv head: the shuffle kernel's shape
  if (dtype != SEMIPD_BF16 && dtype != SEMIPD_F16) return 0;
  return kv_dtype == dtype || kv_dtype == SEMIPD_F8E5M2 || kv_dtype == SEMIPD_F8E4M3;
}

int semipd_decode_rope_attention_planes(void* out, float* attn_logits, const float* planes, int n_planes, int64_t plane_elems,
                                        void* k_buf, void* v_buf, const int64_t* loc, const float* cos_sin_cache,
                                        const int64_t* positions, const int32_t* kv_indptr, const int32_t* kv_indices,
                                        int64_t batch, int num_q_heads, int num_kv_heads, int head_size, int64_t o_stride,
                                        int64_t kbuf_stride, int64_t vbuf_stride, int waves, int zsplits, float sm_scale,
                                        float logit_cap, int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && n_planes >= 1 && head_size > 0 && num_q_heads > 0 && num_kv_heads > 0, SEMIPD_EINVAL,
                   "decode_rope_attention_planes: bad sizes");
  if (batch == 0) return 0;
  SEMIPD_CHECK_ARG(out && planes && k_buf && v_buf && loc && cos_sin_cache && positions && kv_indptr && kv_indices,
                   SEMIPD_EINVAL, "decode_rope_attention_planes: null pointer");
  SEMIPD_CHECK_ARG(semipd_decode_rope_attention_planes_supported(num_q_heads, num_kv_heads, head_size, dtype, kv_dtype),
                   SEMIPD_ESHAPE,
                   "decode_rope_attention_planes: needs Hq %% Hkv == 0, 2 .. 16 q heads per kv head, head size 64 / 128, "
                   "bf16 / f16 activations, pool rows in that type or fp8 (got Hq %d Hkv %d head %d dtype %d kv_dtype %d)",
                   num_q_heads, num_kv_heads, head_size, dtype, kv_dtype);
  SEMIPD_CHECK_ARG(waves == 4 || waves == 8, SEMIPD_EINVAL, "decode_rope_attention_planes: waves (kv splits per workgroup) must be 4 or 8");
  SEMIPD_CHECK_ARG(zsplits >= 1 && zsplits <= 64 && (zsplits == 1 || attn_logits), SEMIPD_EINVAL,
                   "decode_rope_attention_planes: 1 .. 64 workgroups per (request, kv head); attn_logits scratch required above 1");
  SEMIPD_CHECK_ARG(batch * num_kv_heads < (1ll << 31), SEMIPD_EINVAL, "decode_rope_attention_planes: grid too large");
  const int64_t row_elems = (int64_t)(num_q_heads + 2 * num_kv_heads) * head_size;
  SEMIPD_CHECK_ARG(o_stride % 8 == 0 && kbuf_stride % 16 == 0 && vbuf_stride % 16 == 0 && plane_elems % 4 == 0 &&
                       plane_elems >= batch * row_elems && aligned16(out) && aligned16(planes) && aligned16(k_buf) &&
                       aligned16(v_buf),
                   SEMIPD_EALIGN, "decode_rope_attention_planes: 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  int rc = 0;
#define GO(TT, KVT)                                                                                                       \
  rc = launch_decode_rope_attn<TT, KVT>(out, attn_logits, zsplits, planes, n_planes, plane_elems, k_buf, v_buf, loc, cos_sin_cache, positions, \
                                          kv_indptr, kv_indices, batch, num_q_heads, num_kv_heads, head_size, o_stride,    \
                                          kbuf_stride, vbuf_stride, waves, sm_scale, logit_cap, st)
  SEMIPD_DISPATCH_HALF(dtype, T, {
    if (kv_dtype == dtype) GO(T, T);
    else if (kv_dtype == SEMIPD_F8E5M2) GO(T, f8e5m2_t);
    else GO(T, f8e4m3_t);
  });
#undef GO
  if (rc == 0 && zsplits > 1)
    rc = launch_decode_stage2(out, attn_logits, kv_indptr, batch, num_q_heads, head_size, o_stride, zsplits, dtype, st);
  return rc;
}

}  // extern "C"
