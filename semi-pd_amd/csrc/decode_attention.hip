// Paged decode attention for gfx950: split-KV flash-decoding.
//
// Stage 1: one workgroup (4 waves) per (request, kv-head tile, kv split).  A KV row
// (head_dim * 2 bytes, contiguous in the pool row) is read by LPR = head_dim/8 lanes,
// 16 bytes per lane, so one wave-wide load instruction fetches 64/LPR whole rows.
// All G q-heads that share the kv head are processed from the same registers: the
// KV bytes are read from HBM exactly once per kv head (GQA grouping).  The QK dot is
// reduced inside a 16-lane DPP row (no LDS traffic), softmax is online in fp32 per
// lane group, and lane groups / waves are merged once at the end (shuffles + LDS).
// Stage 2 merges the per-split partials with the log-sum-exp trick.
//
// Mirrors decode_attention_fwd (layers/attention/triton_ops/decode_attention.py:625-670).
#include "common.h"

namespace semipd {

template <int CTRL>
__device__ inline float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}

// all-reduce (sum) across the LPR consecutive lanes of a lane group
template <int LPR>
__device__ inline float group_sum(float v) {
  if (LPR >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LPR >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LPR >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if (LPR >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ inline float safe_exp_diff(float a, float b) {
  // exp(a - b) with a <= b; a may be -inf (result 0); b == -inf only if a == -inf (result 1 is
  // harmless because the value it scales is 0)
  return (a == -INFINITY) ? (b == -INFINITY ? 1.f : 0.f) : __expf(a - b);
}

template <typename T, int LPR, int G, int U, typename KV = T>
__global__ void __launch_bounds__(256)
decode_stage1_kernel(T* __restrict__ out, const T* __restrict__ q, const KV* __restrict__ k_buf,
                     const KV* __restrict__ v_buf, const int32_t* __restrict__ kv_indptr,
                     const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
                     int num_q_heads, int num_kv_heads, int group, int tiles_per_kv, int D,
                     int64_t q_stride, int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                     int num_kv_splits, float sm_scale, float logit_cap) {
  constexpr int V = 8;
  constexpr int RPW = 64 / LPR;  // rows per wave per load
  constexpr int STEP = 4 * RPW;  // rows per workgroup per load
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int b = blockIdx.x;
  const int hk = blockIdx.y / tiles_per_kv;
  const int tile = blockIdx.y - hk * tiles_per_kv;
  const int split = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, grp = lane / LPR;
  const bool act = sub * V < D;
  const int h0 = tile * G;                       // first q head (within the group) of this tile
  const int heads = min(G, group - h0);          // valid heads in this tile
  const int hq0 = hk * group + h0;

  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  if (s_end <= s_begin) {  // empty split: stage 2 skips it (decode_attention.py:97)
    if (num_kv_splits == 1) {  // empty sequence and no stage 2: define the output as zeros
      for (int i = threadIdx.x; i < heads * D; i += blockDim.x)
        out[(int64_t)b * o_stride + (int64_t)(hq0 + i / D) * D + (i % D)] = Elem<T>::from_f(0.f);
    }
    return;
  }

  float qf[G][V];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    if (act && h < heads) {
      Vec16<T> a = load16(q + (int64_t)b * q_stride + (int64_t)(hq0 + h) * D + sub * V);
#pragma unroll
      for (int j = 0; j < V; ++j) qf[h][j] = Elem<T>::to_f(a.e[j]) * sm_scale;
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) qf[h][j] = 0.f;
    }
  }
  float m[G], l[G], acc[G][V];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) acc[h][j] = 0.f;
  }

  const int32_t* idx_base = kv_indices + kv_start;
  const int64_t head_off = (int64_t)hk * D + sub * V;
  const int n_iter = (s_end - s_begin + STEP * U - 1) / (STEP * U);
  const int row0 = s_begin + wave * RPW + grp;

  for (int it = 0; it < n_iter; ++it) {
    int tok[U];
    int32_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      tok[u] = row0 + (it * U + u) * STEP;
      idx[u] = tok[u] < s_end ? idx_base[tok[u]] : 0;
    }
    // pool rows in their storage type (fp8: 8 bytes per 8 elements in flight), expanded right before use
    using KVT = KVTraits<T, KV>;
    typename KVT::Raw kraw[U], vraw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      kraw[u] = KVT::zero();
      if (act) kraw[u] = KVT::load8(k_buf + (int64_t)idx[u] * kbuf_stride + head_off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vraw[u] = KVT::zero();
      if (act) vraw[u] = KVT::load8(v_buf + (int64_t)idx[u] * vbuf_stride + head_off);
    }
    Vec16<T> kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint4 ke = KVT::expand(kraw[u]), ve = KVT::expand(vraw[u]);
      kr[u] = *reinterpret_cast<const Vec16<T>*>(&ke);
      vr[u] = *reinterpret_cast<const Vec16<T>*>(&ve);
    }
    float s[U][G];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kf[V];
#pragma unroll
      for (int j = 0; j < V; ++j) kf[j] = act ? Elem<T>::to_f(kr[u].e[j]) : 0.f;
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) d = fmaf(qf[h][j], kf[j], d);
        d = group_sum<LPR>(d);
        if (logit_cap > 0.f) d = logit_cap * tanhf(d / logit_cap);
        s[u][h] = tok[u] < s_end ? d : -INFINITY;
      }
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
      float mx = m[h];
#pragma unroll
      for (int u = 0; u < U; ++u) mx = fmaxf(mx, s[u][h]);
      if (mx > m[h]) {
        const float sc = safe_exp_diff(m[h], mx);
        l[h] *= sc;
#pragma unroll
        for (int j = 0; j < V; ++j) acc[h][j] *= sc;
        m[h] = mx;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float p = safe_exp_diff(s[u][h], mx);
        s[u][h] = p;
        l[h] += p;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float vf[V];
#pragma unroll
      for (int j = 0; j < V; ++j) vf[j] = act ? Elem<T>::to_f(vr[u].e[j]) : 0.f;
#pragma unroll
      for (int h = 0; h < G; ++h) {
#pragma unroll
        for (int j = 0; j < V; ++j) acc[h][j] = fmaf(s[u][h], vf[j], acc[h][j]);
      }
    }
  }

  // merge the RPW lane groups of this wave
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      const float mo = __shfl_xor(m[h], off, 64);
      const float lo = __shfl_xor(l[h], off, 64);
      const float mn = fmaxf(m[h], mo);
      const float sa = safe_exp_diff(m[h], mn), sb = safe_exp_diff(mo, mn);
      l[h] = l[h] * sa + lo * sb;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float ao = __shfl_xor(acc[h][j], off, 64);
        acc[h][j] = acc[h][j] * sa + ao * sb;
      }
      m[h] = mn;
    }
  }
  // per-wave results -> LDS: [wave][h][0..D) acc, then m, l
  const int hstride = D + 2;
  float* my = smem + (wave * G) * hstride;
  if (grp == 0) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      if (act) {
#pragma unroll
        for (int j = 0; j < V; ++j) my[h * hstride + sub * V + j] = acc[h][j];
      }
      if (sub == 0) {
        my[h * hstride + D] = m[h];
        my[h * hstride + D + 1] = l[h];
      }
    }
  }
  __syncthreads();
  const int Dv = D;
  for (int i = threadIdx.x; i < heads * D; i += blockDim.x) {
    const int h = i / D, d = i - h * D;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, smem[(w * G + h) * hstride + D]);
    float ll = 0.f, aa = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* p = smem + (w * G + h) * hstride;
      const float sc = safe_exp_diff(p[D], mm);
      ll += p[D + 1] * sc;
      aa += p[d] * sc;
    }
    const float o = aa / ll;
    const int hq = hq0 + h;
    if (num_kv_splits == 1) {
      out[(int64_t)b * o_stride + (int64_t)hq * Dv + d] = Elem<T>::from_f(o);
    } else {
      float* dst = attn_logits + (((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (Dv + 1);
      dst[d] = o;
      if (d == 0) dst[Dv] = mm + __logf(ll);
    }
  }
}

// Generic fallback: any Dk / Dv (odd sizes, MLA 576/512 until the dedicated kernel takes it).
// One wave per (request, q head, split); lanes stride over the head dim.
template <typename T>
__global__ void __launch_bounds__(64)
decode_stage1_generic_kernel(T* __restrict__ out, const T* __restrict__ q,
                             const T* __restrict__ k_buf, const T* __restrict__ v_buf,
                             const int32_t* __restrict__ kv_indptr,
                             const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
                             int num_q_heads, int group, int Dk, int Dv, int64_t q_stride,
                             int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                             int num_kv_splits, float sm_scale, float logit_cap) {
  constexpr int MAXR = 9;  // up to 576 dims
  const int b = blockIdx.x, hq = blockIdx.y, split = blockIdx.z;
  const int hk = hq / group;
  const int lane = threadIdx.x;
  const int kv_start = kv_indptr[b];
  const int seq_len = kv_indptr[b + 1] - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  if (s_end <= s_begin) return;
  float qf[MAXR], acc[MAXR];
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int d = lane + r * 64;
    qf[r] = d < Dk ? Elem<T>::to_f(q[(int64_t)b * q_stride + (int64_t)hq * Dk + d]) * sm_scale : 0.f;
    acc[r] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int t = s_begin; t < s_end; ++t) {
    const int64_t idx = kv_indices[kv_start + t];
    const T* kr = k_buf + idx * kbuf_stride + (int64_t)hk * Dk;
    const T* vr = v_buf + idx * vbuf_stride + (int64_t)hk * Dv;
    float d = 0.f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int dd = lane + r * 64;
      if (dd < Dk) d = fmaf(qf[r], Elem<T>::to_f(kr[dd]), d);
    }
    d = wave_sum(d);
    if (logit_cap > 0.f) d = logit_cap * tanhf(d / logit_cap);
    const float mn = fmaxf(m, d);
    const float sc = safe_exp_diff(m, mn);
    const float p = __expf(d - mn);
    l = l * sc + p;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int dd = lane + r * 64;
      const float vv = dd < Dv ? Elem<T>::to_f(vr[dd]) : 0.f;
      acc[r] = acc[r] * sc + p * vv;
    }
    m = mn;
  }
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int dd = lane + r * 64;
    if (dd < Dv) {
      const float o = acc[r] / l;
      if (num_kv_splits == 1) {
        out[(int64_t)b * o_stride + (int64_t)hq * Dv + dd] = Elem<T>::from_f(o);
      } else {
        attn_logits[(((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (Dv + 1) + dd] = o;
      }
    }
  }
  if (lane == 0 && num_kv_splits > 1)
    attn_logits[(((int64_t)b * num_q_heads + hq) * num_kv_splits + split) * (Dv + 1) + Dv] =
        m + __logf(l);
}

// Stage 2 (decode_attention.py:476-531): merge split partials.
// The merge of one (request, head): weights w_s = exp(lse_s - max lse), out = sum_s w_s o_s / sum_s w_s, sums in split order.
// Up to 64 splits the log-sum-exps are fetched by one load per lane and the partial rows eight splits at a time, so the
// kernel waits for two or three memory latencies instead of one per split.
template <typename T>
__device__ __forceinline__ void stage2_merge(T* __restrict__ out_row, const float* __restrict__ base, int n_valid, int Dv,
                                             int first_d, int step_d) {
  const int lane = threadIdx.x & 63;
  float e_max = -INFINITY, e_sum = 0.f;
  if (n_valid <= 64) {
    const float mine = lane < n_valid ? base[(int64_t)lane * (Dv + 1) + Dv] : -INFINITY;
    float mx = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    e_max = mx;
    const float w = lane < n_valid ? __expf(mine - e_max) : 0.f;
    for (int s = 0; s < n_valid; ++s) e_sum += __shfl(w, s, 64);          // split order, like the loop it replaces
    const float inv = e_sum > 0.f ? 1.f / e_sum : 0.f;
    // (every lane runs every iteration: the shuffles below read the weights from lanes 0 .. n_valid - 1)
    for (int d0 = 0; d0 < Dv; d0 += step_d) {
      const bool ok = d0 + first_d < Dv;
      const int d = ok ? d0 + first_d : 0;
      float acc = 0.f;
      int s = 0;
      for (; s + 8 <= n_valid; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(s + u) * (Dv + 1) + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(__shfl(w, s + u, 64), v[u], acc);   // pinned: left to itself the
                                       // compiler multiplies pairs (v_pk_mul) and adds here but fuses in the tail below
      }
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s + u < n_valid) v[u] = base[(int64_t)(s + u) * (Dv + 1) + d];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s + u < n_valid) acc = __builtin_fmaf(__shfl(w, s + u, 64), v[u], acc);
      if (ok) out_row[d] = Elem<T>::from_f(acc * inv);
    }
    return;
  }
  // more than 64 splits: two passes so that the loads of the second one are independent of each other
  for (int s = 0; s < n_valid; ++s) e_max = fmaxf(e_max, base[(int64_t)s * (Dv + 1) + Dv]);
  for (int s = 0; s < n_valid; ++s) e_sum += __expf(base[(int64_t)s * (Dv + 1) + Dv] - e_max);
  const float inv = e_sum > 0.f ? 1.f / e_sum : 0.f;
  for (int d = first_d; d < Dv; d += step_d) {
    float acc = 0.f;
#pragma unroll 4
    for (int s = 0; s < n_valid; ++s)
      acc += __expf(base[(int64_t)s * (Dv + 1) + Dv] - e_max) * base[(int64_t)s * (Dv + 1) + d];
    out_row[d] = Elem<T>::from_f(acc * inv);
  }
}

template <typename T>
__global__ void decode_stage2_kernel(T* __restrict__ out, const float* __restrict__ attn_logits,
                                     const int32_t* __restrict__ kv_indptr, int num_q_heads, int Dv,
                                     int64_t o_stride, int num_kv_splits) {
  const int b = blockIdx.x, hq = blockIdx.y;
  const int seq_len = kv_indptr[b + 1] - kv_indptr[b];
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const float* base = attn_logits + ((int64_t)b * num_q_heads + hq) * num_kv_splits * (Dv + 1);
  // splits [0, n_valid) are non-empty
  const int n_valid = per_split > 0 ? min(num_kv_splits, (seq_len + per_split - 1) / per_split) : 0;
  stage2_merge<T>(out + (int64_t)b * o_stride + (int64_t)hq * Dv, base, n_valid, Dv, threadIdx.x, blockDim.x);
}

// The same merge with one WAVE per head and four heads per workgroup: 128 heads x 128 requests of MLA decode are 16 k
// workgroups of a few KB each in the kernel above (38 us for 67 MB); this form is a quarter of the workgroups.
template <typename T>
__global__ void __launch_bounds__(256)
decode_stage2_wave_kernel(T* __restrict__ out, const float* __restrict__ attn_logits, const int32_t* __restrict__ kv_indptr,
                          int num_q_heads, int Dv, int64_t o_stride, int num_kv_splits) {
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  const int hq = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (hq >= num_q_heads) return;
  const int seq_len = kv_indptr[b + 1] - kv_indptr[b];
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const float* base = attn_logits + ((int64_t)b * num_q_heads + hq) * num_kv_splits * (Dv + 1);
  const int n_valid = per_split > 0 ? min(num_kv_splits, (seq_len + per_split - 1) / per_split) : 0;
  stage2_merge<T>(out + (int64_t)b * o_stride + (int64_t)hq * Dv, base, n_valid, Dv, lane, 64);
}

template <typename T>
static int launch_decode_stage2_t(T* out, const float* attn_logits, const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                                  int head_dim_v, int64_t o_stride, int num_kv_splits, hipStream_t st) {
  if (head_dim_v >= 256 && batch * num_q_heads >= 4096) {
    dim3 grid((unsigned)batch, (unsigned)((num_q_heads + 3) / 4));
    hipLaunchKernelGGL((decode_stage2_wave_kernel<T>), grid, dim3(256), 0, st, out, attn_logits, kv_indptr,
                       num_q_heads, head_dim_v, o_stride, num_kv_splits);
  } else {
    dim3 grid((unsigned)batch, (unsigned)num_q_heads);
    const int threads = head_dim_v <= 64 ? 64 : head_dim_v <= 128 ? 128 : 256;
    hipLaunchKernelGGL((decode_stage2_kernel<T>), grid, dim3(threads), 0, st, out, attn_logits, kv_indptr, num_q_heads,
                       head_dim_v, o_stride, num_kv_splits);
  }
  return launch_status("decode_stage2");
}

template <typename T, int LPR, typename KV = T>
static int launch_stage1(T* out, const T* q, const KV* k_buf, const KV* v_buf, const int32_t* kv_indptr,
                         const int32_t* kv_indices, float* attn_logits, int64_t batch, int Hq, int Hkv,
                         int D, int64_t q_stride, int64_t o_stride, int64_t kbuf_stride,
                         int64_t vbuf_stride, int splits, float sm_scale, float logit_cap,
                         hipStream_t st) {
  const int group = Hq / Hkv;
  int G = group >= 8 ? 8 : group > 2 ? 4 : group;  // heads per workgroup tile: 1, 2, 4, 8
  if (group == 3) G = 4;
  const int tiles = (group + G - 1) / G;
  dim3 grid((unsigned)batch, (unsigned)(Hkv * tiles), (unsigned)splits), block(256);
  const size_t lds = (size_t)4 * G * (D + 2) * sizeof(float);
#define S1(GG, UU)                                                                              \
  hipLaunchKernelGGL((decode_stage1_kernel<T, LPR, GG, UU, KV>), grid, block, lds, st, out, q, k_buf, \
                     v_buf, kv_indptr, kv_indices, attn_logits, Hq, Hkv, group, tiles, D, q_stride, \
                     o_stride, kbuf_stride, vbuf_stride, splits, sm_scale, logit_cap)
  switch (G) {
    case 1: S1(1, 4); break;
    case 2: S1(2, 4); break;
    case 4: S1(4, 4); break;
    default: S1(8, 2); break;
  }
#undef S1
  return launch_status("decode_stage1");
}

// defined in decode_attention_mfma.hip
template <typename T, typename KV>
int launch_decode_mfma(T* out, const T* q, const KV* k_buf, const KV* v_buf, const int32_t* kv_indptr,
                       const int32_t* kv_indices, float* attn_logits, int64_t batch, int Hq, int Hkv, int D,
                       int64_t q_stride, int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                       int splits, float sm_scale, float logit_cap, hipStream_t st);

// defined in mla_decode_attention.hip
template <typename T, typename KV>
int launch_mla_decode(T* out, const T* q, const KV* kv_buf, const int32_t* kv_indptr, const int32_t* kv_indices,
                      float* attn_logits, int64_t batch, int Hq, int64_t q_stride, int64_t o_stride,
                      int64_t kvbuf_stride, int splits, float sm_scale, float logit_cap, hipStream_t st);

template <typename T>
static int run_decode(void* out, const void* q, const void* k_buf, const void* v_buf,
                      const int32_t* kv_indptr, const int32_t* kv_indices, float* attn_logits,
                      int64_t batch, int num_q_heads, int num_kv_heads, int head_dim_k,
                      int head_dim_v, int64_t q_stride, int64_t o_stride, int64_t kbuf_stride,
                      int64_t vbuf_stride, int num_kv_splits, float sm_scale, float logit_cap,
                      int dtype, int kv_dtype, hipStream_t st) {
  const bool kv_f8 = kv_dtype != dtype;
  const bool fast = head_dim_k == head_dim_v && head_dim_k % 8 == 0 && head_dim_k <= 256 &&
                    aligned16(q) && aligned16(k_buf) && aligned16(v_buf) && q_stride % 8 == 0 &&
                    kbuf_stride % 8 == 0 && vbuf_stride % 8 == 0;
  int rc = 0;
  const int group = num_q_heads / num_kv_heads;
  const bool mfma = fast && group >= 2 &&
                    (head_dim_k == 64 || head_dim_k == 96 || head_dim_k == 128) &&
                    o_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  const bool mla = head_dim_k == 576 && head_dim_v == 512 && num_kv_heads == 1 && k_buf == v_buf &&
                   kbuf_stride == vbuf_stride && kbuf_stride % 8 == 0 && q_stride % 8 == 0 && aligned16(q) &&
                   aligned16(k_buf) && o_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  if (kv_f8) {
    // fp8 pool rows (mem_cache/memory_pool.py:205-209): every vectorised kernel keeps 8-byte fragments in flight
    // and expands them right before use; only the scalar fallback (ragged head sizes) reads activation-type rows
    SEMIPD_CHECK_ARG(kv_dtype == SEMIPD_F8E5M2 || kv_dtype == SEMIPD_F8E4M3, SEMIPD_EDTYPE,
                     "decode_attention: unsupported kv_dtype %d", kv_dtype);
    SEMIPD_CHECK_ARG(mfma || mla || fast, SEMIPD_ESHAPE,
                     "decode_attention: fp8 KV rows need 8-element aligned heads of equal K / V width (or MLA rows)");
    if (!mfma && !mla) {
      // MHA (one query head per kv head) and head sizes without an MFMA instantiation
      const int D = head_dim_k;
#define S1F8(LPRV, KVT_)                                                                                           \
  rc = launch_stage1<T, LPRV, KVT_>((T*)out, (const T*)q, (const KVT_*)k_buf, (const KVT_*)v_buf, kv_indptr,        \
                                    kv_indices, attn_logits, batch, num_q_heads, num_kv_heads, D, q_stride, o_stride, \
                                    kbuf_stride, vbuf_stride, num_kv_splits, sm_scale, logit_cap, st)
      if (kv_dtype == SEMIPD_F8E5M2) {
        if (D <= 64) S1F8(8, f8e5m2_t);
        else if (D <= 128) S1F8(16, f8e5m2_t);
        else S1F8(32, f8e5m2_t);
      } else {
        if (D <= 64) S1F8(8, f8e4m3_t);
        else if (D <= 128) S1F8(16, f8e4m3_t);
        else S1F8(32, f8e4m3_t);
      }
#undef S1F8
    } else if (mla && kv_dtype == SEMIPD_F8E5M2)
      rc = launch_mla_decode<T, f8e5m2_t>((T*)out, (const T*)q, (const f8e5m2_t*)k_buf, kv_indptr, kv_indices, attn_logits,
                                          batch, num_q_heads, q_stride, o_stride, kbuf_stride, num_kv_splits, sm_scale,
                                          logit_cap, st);
    else if (mla)
      rc = launch_mla_decode<T, f8e4m3_t>((T*)out, (const T*)q, (const f8e4m3_t*)k_buf, kv_indptr, kv_indices, attn_logits,
                                          batch, num_q_heads, q_stride, o_stride, kbuf_stride, num_kv_splits, sm_scale,
                                          logit_cap, st);
    else if (kv_dtype == SEMIPD_F8E5M2)
      rc = launch_decode_mfma<T, f8e5m2_t>((T*)out, (const T*)q, (const f8e5m2_t*)k_buf, (const f8e5m2_t*)v_buf,
                                           kv_indptr, kv_indices, attn_logits, batch, num_q_heads, num_kv_heads,
                                           head_dim_k, q_stride, o_stride, kbuf_stride, vbuf_stride, num_kv_splits,
                                           sm_scale, logit_cap, st);
    else
      rc = launch_decode_mfma<T, f8e4m3_t>((T*)out, (const T*)q, (const f8e4m3_t*)k_buf, (const f8e4m3_t*)v_buf,
                                           kv_indptr, kv_indices, attn_logits, batch, num_q_heads, num_kv_heads,
                                           head_dim_k, q_stride, o_stride, kbuf_stride, vbuf_stride, num_kv_splits,
                                           sm_scale, logit_cap, st);
  } else if (mla) {
    // DeepSeek latent rows shared by all heads (mla_decode_attention.hip)
    rc = launch_mla_decode<T, T>((T*)out, (const T*)q, (const T*)k_buf, kv_indptr, kv_indices, attn_logits, batch,
                              num_q_heads, q_stride, o_stride, kbuf_stride, num_kv_splits, sm_scale, logit_cap, st);
  } else if (mfma) {
    // GQA / MQA: matrix-core kernel (decode_attention_mfma.hip)
    rc = launch_decode_mfma<T, T>((T*)out, (const T*)q, (const T*)k_buf, (const T*)v_buf, kv_indptr, kv_indices,
                               attn_logits, batch, num_q_heads, num_kv_heads, head_dim_k, q_stride, o_stride,
                               kbuf_stride, vbuf_stride, num_kv_splits, sm_scale, logit_cap, st);
  } else if (fast) {
    const int D = head_dim_k;
    if (D <= 64)
      rc = launch_stage1<T, 8>((T*)out, (const T*)q, (const T*)k_buf, (const T*)v_buf, kv_indptr,
                               kv_indices, attn_logits, batch, num_q_heads, num_kv_heads, D, q_stride,
                               o_stride, kbuf_stride, vbuf_stride, num_kv_splits, sm_scale, logit_cap,
                               st);
    else if (D <= 128)
      rc = launch_stage1<T, 16>((T*)out, (const T*)q, (const T*)k_buf, (const T*)v_buf, kv_indptr,
                                kv_indices, attn_logits, batch, num_q_heads, num_kv_heads, D,
                                q_stride, o_stride, kbuf_stride, vbuf_stride, num_kv_splits, sm_scale,
                                logit_cap, st);
    else
      rc = launch_stage1<T, 32>((T*)out, (const T*)q, (const T*)k_buf, (const T*)v_buf, kv_indptr,
                                kv_indices, attn_logits, batch, num_q_heads, num_kv_heads, D,
                                q_stride, o_stride, kbuf_stride, vbuf_stride, num_kv_splits, sm_scale,
                                logit_cap, st);
  } else {
    dim3 grid((unsigned)batch, (unsigned)num_q_heads, (unsigned)num_kv_splits);
    hipLaunchKernelGGL((decode_stage1_generic_kernel<T>), grid, dim3(64), 0, st, (T*)out, (const T*)q,
                       (const T*)k_buf, (const T*)v_buf, kv_indptr, kv_indices, attn_logits,
                       num_q_heads, num_q_heads / num_kv_heads, head_dim_k, head_dim_v, q_stride,
                       o_stride, kbuf_stride, vbuf_stride, num_kv_splits, sm_scale, logit_cap);
    rc = launch_status("decode_stage1_generic");
  }
  if (rc == 0 && num_kv_splits > 1)
    rc = launch_decode_stage2_t<T>((T*)out, attn_logits, kv_indptr, batch, num_q_heads, head_dim_v, o_stride, num_kv_splits, st);
  return rc;
}

// stage 2 for the one-launch decode form with several workgroups per (request, kv head) (decode_attention_fused.hip)
int launch_decode_stage2(void* out, const float* attn_logits, const int32_t* kv_indptr, int64_t batch, int num_q_heads,
                         int head_dim_v, int64_t o_stride, int num_kv_splits, int dtype, hipStream_t st) {
  SEMIPD_DISPATCH_HALF(dtype, T, return launch_decode_stage2_t<T>((T*)out, attn_logits, kv_indptr, batch, num_q_heads, head_dim_v, o_stride, num_kv_splits, st));
  return 0;
}

}  // namespace semipd

using namespace semipd;

extern "C" int semipd_decode_attention(void* out, const void* q, const void* k_buf, const void* v_buf,
                                       const int32_t* kv_indptr, const int32_t* kv_indices,
                                       float* attn_logits, int64_t batch, int num_q_heads,
                                       int num_kv_heads, int head_dim_k, int head_dim_v,
                                       int64_t q_stride, int64_t o_stride, int64_t kbuf_stride,
                                       int64_t vbuf_stride, int num_kv_splits, float sm_scale,
                                       float logit_cap, int dtype, int kv_dtype, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && num_q_heads > 0 && num_kv_heads > 0 && head_dim_k > 0 &&
                       head_dim_v > 0 && num_kv_splits > 0,
                   SEMIPD_EINVAL, "decode_attention: bad sizes");
  SEMIPD_CHECK_ARG(num_q_heads % num_kv_heads == 0, SEMIPD_ESHAPE,
                   "decode_attention: Hq %d not a multiple of Hkv %d", num_q_heads, num_kv_heads);
  SEMIPD_CHECK_ARG(head_dim_k <= 576 && head_dim_v <= 576, SEMIPD_ESHAPE,
                   "decode_attention: head dims up to 576 supported");
  SEMIPD_CHECK_ARG(batch < 65536 * 16 && num_kv_splits <= 65535, SEMIPD_EINVAL,
                   "decode_attention: grid too large");
  if (batch == 0) return 0;
  SEMIPD_CHECK_ARG(out && q && k_buf && v_buf && kv_indptr && kv_indices, SEMIPD_EINVAL,
                   "decode_attention: null pointer");
  SEMIPD_CHECK_ARG(num_kv_splits == 1 || attn_logits, SEMIPD_EINVAL,
                   "decode_attention: attn_logits scratch required when num_kv_splits > 1");
  SEMIPD_DISPATCH_HALF(dtype, T, return run_decode<T>(out, q, k_buf, v_buf, kv_indptr, kv_indices, attn_logits, batch, num_q_heads, num_kv_heads, head_dim_k, head_dim_v, q_stride, o_stride, kbuf_stride, vbuf_stride, num_kv_splits, sm_scale, logit_cap, dtype, kv_dtype, as_stream(stream)));
  return 0;
}
