// Stochastic sampling over fp32 probability rows [B, V] for gfx950.
//
// Reference behaviour: layers/sampler.py:77-136 (logits / T -> softmax -> top-k / top-p / min-p sampling)
// on top of the sgl-kernel ops top_k_top_p_sampling_from_probs / min_p_sampling_from_probs /
// top_k_renorm_prob / top_p_renorm_prob (sgl-kernel/csrc/torch_extension.cc:151-175, arithmetic from
// FlashInfer 0.2.3 sampling.cuh, which is not vendored in the reference tree; pinned by
// sgl-kernel/tests/test_sampling.py:8-141).
//
// One 1024-thread workgroup per row; a 128 K-entry row (512 KB) is re-read from L2 by every pass.
//   * sampling is inverse-CDF in token-index order: each of the 16 waves owns a contiguous segment,
//     segment masses are combined in LDS, and only the wave whose segment holds the target walks it
//     with a 64-lane prefix scan per 256-element sub-tile;
//   * top-k/top-p "joint" filtering is the rejection scheme of the reference kernel: sample among
//     probs > pivot, accept iff fewer than k tokens and less than p mass lie strictly above the
//     sampled token, otherwise raise the pivot to the sampled probability (<= 32 rounds);
//   * the renorm ops find their pivot by bisection on the float bit pattern (3 thresholds per pass).
#include "common.h"

namespace semipd {

constexpr int kST = 1024;          // threads per row
constexpr int kSW = kST / 64;      // waves per row

struct SampLds {
  float f[3][kSW];
  float seg[kSW];
  int res_idx;
  float res_val;
};

template <bool VEC, typename F>
__device__ inline void visit_row(const float* __restrict__ row, int V, F f) {
  if (VEC) {
    for (int i = threadIdx.x * 4; i < V; i += kST * 4) {
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      f(v.x, i);
      f(v.y, i + 1);
      f(v.z, i + 2);
      f(v.w, i + 3);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += kST) f(row[i], i);
  }
}

// block-wide sums of up to three values; every thread gets the results
template <int N>
__device__ inline void block_sum_n(float (&v)[N], SampLds& s) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < N; ++j) s.f[j][wid] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kSW; ++w) t += s.f[j][w];
    v[j] = t;
  }
}

__device__ inline float block_max_f(float v, SampLds& s) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) s.f[0][wid] = v;
  __syncthreads();
  float t = s.f[0][0];
#pragma unroll
  for (int w = 1; w < kSW; ++w) t = fmaxf(t, s.f[0][w]);
  return t;
}

template <bool GE> __device__ inline bool keep(float x, float pivot) { return GE ? (x >= pivot) : (x > pivot); }

// Inverse-CDF sample among {i : keep(row[i], pivot)} with probability proportional to row[i], using
// the uniform number u in [0, 1).  Returns the index (every thread gets it) and the kept mass in *mass;
// -1 if nothing is kept.
template <bool VEC, bool GE>
__device__ inline int sample_kept(const float* __restrict__ row, int V, float pivot, float u, SampLds& s,
                                  float* mass) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int seg_len = ((V + kSW * 256 - 1) / (kSW * 256)) * 256;
  const int seg_lo = wid * seg_len;
  const int seg_hi = min(V, seg_lo + seg_len);
  // ---- step 1: mass of every wave segment ----
  float local = 0.f;
  for (int b = seg_lo + lane * 4; b < seg_hi; b += 256) {
    float x[4];
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(row + b);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (b + e < seg_hi) ? row[b + e] : -1.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) local += (x[e] > 0.f && keep<GE>(x[e], pivot)) ? x[e] : 0.f;
  }
  local = wave_sum(local);
  __syncthreads();
  if (lane == 0) s.seg[wid] = local;
  if (threadIdx.x == 0) s.res_idx = -1;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int w = 0; w < kSW; ++w) total += s.seg[w];
  *mass = total;
  if (!(total > 0.f)) return -1;
  const float target = u * total;
  int pick = -1;
  float before = 0.f, cum = 0.f;
#pragma unroll
  for (int w = 0; w < kSW; ++w) {
    const float sw = s.seg[w];
    if (sw > 0.f && (pick < 0 || target >= cum)) {  // last non-empty segment whose start is <= target
      pick = w;
      before = cum;
    }
    cum += sw;
  }
  // ---- step 2: the owning wave walks its segment ----
  if (wid == pick) {
    float rem = target - before;
    int found = -1, last_kept = -1;
    for (int b0 = seg_lo; b0 < seg_hi && found < 0; b0 += 256) {
      const int b = b0 + lane * 4;
      float x[4];
      if (VEC) {
        if (b < seg_hi) {
          const float4 v = *reinterpret_cast<const float4*>(row + b);
          x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
          x[0] = x[1] = x[2] = x[3] = -1.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (b + e < seg_hi) ? row[b + e] : -1.f;
      }
      float a[4], lane_sum = 0.f;
      int lane_last = -1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool k = x[e] > 0.f && keep<GE>(x[e], pivot);
        a[e] = k ? x[e] : 0.f;
        lane_sum += a[e];
        if (k) lane_last = b + e;
      }
      float incl = lane_sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      const float tile_total = __shfl(incl, 63, 64);
      const unsigned long long hit = __ballot(incl > rem);
      if (hit != 0ull) {
        const int L = __builtin_ctzll(hit);
        if (lane == L) {
          float c = incl - lane_sum;
          int idx = lane_last;
#pragma unroll
          for (int e = 3; e >= 0; --e) {  // first e whose running sum exceeds rem
            float ce = c;
            for (int j = 0; j <= e; ++j) ce += a[j];
            if (a[e] > 0.f && ce > rem) idx = b + e;
          }
          s.res_idx = idx;
        }
        found = 1;
      } else {
        rem -= tile_total;
        int ll = lane_last;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ll = max(ll, __shfl_xor(ll, o, 64));
        last_kept = max(last_kept, ll);
      }
    }
    if (found < 0 && lane == 0) s.res_idx = last_kept;  // rounding pushed the target past the end
  }
  __syncthreads();
  return s.res_idx;
}

// ------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(kST)
softmax_temperature_kernel(float* __restrict__ logits, const float* __restrict__ temperatures, int V) {
  __shared__ SampLds s;
  float* row = logits + (int64_t)blockIdx.x * V;
  const float t = temperatures ? temperatures[blockIdx.x] : 1.f;
  float m = -INFINITY;
  visit_row<VEC>(row, V, [&](float x, int) { m = fmaxf(m, x / t); });
  m = block_max_f(m, s);
  float acc[1] = {0.f};
  visit_row<VEC>(row, V, [&](float x, int) { acc[0] += expf(x / t - m); });
  block_sum_n<1>(acc, s);
  const float inv = 1.f / acc[0];
  if (VEC) {
    for (int i = threadIdx.x * 4; i < V; i += kST * 4) {
      float4 v = *reinterpret_cast<const float4*>(row + i);
      v.x = expf(v.x / t - m) * inv;
      v.y = expf(v.y / t - m) * inv;
      v.z = expf(v.z / t - m) * inv;
      v.w = expf(v.w / t - m) * inv;
      *reinterpret_cast<float4*>(row + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < V; i += kST) row[i] = expf(row[i] / t - m) * inv;
  }
}

template <bool VEC>
__global__ void __launch_bounds__(kST)
top_k_top_p_sample_kernel(const float* __restrict__ probs, const float* __restrict__ uniform,
                          const int32_t* __restrict__ top_ks, int top_k_val, const float* __restrict__ top_ps,
                          float top_p_val, int32_t* __restrict__ out, uint8_t* __restrict__ success, int B,
                          int V, int rounds) {
  __shared__ SampLds s;
  const int r = blockIdx.x;
  const float* row = probs + (int64_t)r * V;
  const float k = (float)(top_ks ? top_ks[r] : top_k_val);
  const float p = top_ps ? top_ps[r] : top_p_val;
  float pivot = 0.f;
  int idx = -1, ok = 0;
  for (int round = 0; round < rounds; ++round) {
    float mass;
    const int cand = sample_kept<VEC, false>(row, V, pivot, uniform[(int64_t)round * B + r], s, &mass);
    if (cand < 0) break;
    idx = cand;
    const float pt = row[idx];
    float agg[2] = {0.f, 0.f};  // count and mass strictly above the sampled token
    visit_row<VEC>(row, V, [&](float x, int) {
      if (x > pt) {
        agg[0] += 1.f;
        agg[1] += x;
      }
    });
    block_sum_n<2>(agg, s);
    if (agg[0] < k && agg[1] < p) {
      ok = 1;
      break;
    }
    pivot = pt;
  }
  if (threadIdx.x == 0) {
    out[r] = idx < 0 ? 0 : idx;
    if (success) success[r] = (uint8_t)ok;
  }
}

template <bool VEC>
__global__ void __launch_bounds__(kST)
min_p_sample_kernel(const float* __restrict__ probs, const float* __restrict__ uniform,
                    const float* __restrict__ min_ps, float min_p_val, int32_t* __restrict__ out, int V) {
  __shared__ SampLds s;
  const int r = blockIdx.x;
  const float* row = probs + (int64_t)r * V;
  float m = 0.f;
  visit_row<VEC>(row, V, [&](float x, int) { m = fmaxf(m, x); });
  m = block_max_f(m, s);
  const float pivot = m * (min_ps ? min_ps[r] : min_p_val);
  float mass;
  const int idx = sample_kept<VEC, true>(row, V, pivot, uniform[r], s, &mass);
  if (threadIdx.x == 0) out[r] = idx < 0 ? 0 : idx;
}

// BY_SUM = false: keep the k largest (ties at the k-th value included); true: keep the smallest set of
// largest probabilities whose mass reaches p.  out = kept / sum(kept), 0 elsewhere.
template <bool VEC, bool BY_SUM>
__global__ void __launch_bounds__(kST)
renorm_kernel(const float* __restrict__ probs, float* __restrict__ out, const int32_t* __restrict__ ks,
              int k_val, const float* __restrict__ ps, float p_val, int V) {
  __shared__ SampLds s;
  const int r = blockIdx.x;
  const float* row = probs + (int64_t)r * V;
  float* orow = out + (int64_t)r * V;
  const float target = BY_SUM ? (ps ? ps[r] : p_val) : (float)(ks ? ks[r] : k_val);
  float m = 0.f;
  visit_row<VEC>(row, V, [&](float x, int) { m = fmaxf(m, x); });
  m = block_max_f(m, s);
  // invariant: metric(lo) >= target (metric(0) = everything), metric(hi) < target or hi beyond the max
  uint32_t lo = 0u, hi = __float_as_uint(m) + 1u;
  while (hi - lo > 1u) {
    const uint32_t d = hi - lo;
    const uint32_t mid[3] = {lo + d / 4u, lo + d / 2u, lo + (d / 4u) * 3u};
    const float th[3] = {__uint_as_float(mid[0]), __uint_as_float(mid[1]), __uint_as_float(mid[2])};
    float acc[3] = {0.f, 0.f, 0.f};
    visit_row<VEC>(row, V, [&](float x, int) {
      const float w = BY_SUM ? x : 1.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] += (x >= th[j]) ? w : 0.f;
    });
    block_sum_n<3>(acc, s);
    uint32_t nlo = lo, nhi = hi;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (mid[j] <= lo || mid[j] >= hi) continue;
      if (acc[j] >= target) nlo = max(nlo, mid[j]);
      else nhi = min(nhi, mid[j]);
    }
    if (nhi <= nlo) nhi = nlo + 1u;  // non-monotone rounding of the fp32 sums: stop at nlo
    lo = nlo;
    hi = nhi;
  }
  const float pivot = __uint_as_float(lo);
  float tot[1] = {0.f};
  visit_row<VEC>(row, V, [&](float x, int) { tot[0] += (x >= pivot) ? x : 0.f; });
  block_sum_n<1>(tot, s);
  const float inv = tot[0] > 0.f ? 1.f / tot[0] : 0.f;
  if (VEC) {
    for (int i = threadIdx.x * 4; i < V; i += kST * 4) {
      float4 v = *reinterpret_cast<const float4*>(row + i);
      v.x = v.x >= pivot ? v.x * inv : 0.f;
      v.y = v.y >= pivot ? v.y * inv : 0.f;
      v.z = v.z >= pivot ? v.z * inv : 0.f;
      v.w = v.w >= pivot ? v.w * inv : 0.f;
      *reinterpret_cast<float4*>(orow + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < V; i += kST) {
      const float x = row[i];
      orow[i] = x >= pivot ? x * inv : 0.f;
    }
  }
}

// out[b] = log_softmax(logits[b])[ids[b]] and lse[b] = logsumexp(logits[b])  (Sampler.forward with
// return_logprob on a greedy batch: layers/sampler.py:74-75, 139-155); the top-k logprobs are
// topk(logits) - lse on the host side of the op.
template <bool VEC>
__global__ void __launch_bounds__(kST)
token_logprob_kernel(const float* __restrict__ logits, const int32_t* __restrict__ ids, float* __restrict__ out,
                     float* __restrict__ lse, int V) {
  __shared__ SampLds s;
  const float* row = logits + (int64_t)blockIdx.x * V;
  float m = -INFINITY;
  visit_row<VEC>(row, V, [&](float x, int) { m = fmaxf(m, x); });
  m = block_max_f(m, s);
  float acc[1] = {0.f};
  visit_row<VEC>(row, V, [&](float x, int) { acc[0] += expf(x - m); });
  block_sum_n<1>(acc, s);
  if (threadIdx.x == 0) {
    const float l = m + logf(acc[0]);
    if (lse) lse[blockIdx.x] = l;
    const int id = ids[blockIdx.x];
    out[blockIdx.x] = (id >= 0 && id < V) ? row[id] - l : -INFINITY;
  }
}

static bool rows_vec_ok(const void* p, int64_t V) { return V % 4 == 0 && aligned16(p); }

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_softmax_temperature(float* logits, const float* temperatures, int64_t batch, int64_t vocab,
                               void* stream) {
  SEMIPD_CHECK_ARG(logits && batch >= 0 && vocab > 0 && vocab < (1ll << 30), SEMIPD_EINVAL,
                   "softmax_temperature: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(logits, vocab))
    hipLaunchKernelGGL((softmax_temperature_kernel<true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream),
                       logits, temperatures, (int)vocab);
  else
    hipLaunchKernelGGL((softmax_temperature_kernel<false>), dim3((unsigned)batch), dim3(kST), 0,
                       as_stream(stream), logits, temperatures, (int)vocab);
  return launch_status("softmax_temperature");
}

int semipd_top_k_top_p_sampling_from_probs(const float* probs, const float* uniform_samples,
                                           const int32_t* top_ks, int32_t top_k_val, const float* top_ps,
                                           float top_p_val, int32_t* out_ids, uint8_t* success, int64_t batch,
                                           int64_t vocab, int rounds, void* stream) {
  SEMIPD_CHECK_ARG(probs && uniform_samples && out_ids && batch >= 0 && vocab > 0 && vocab < (1ll << 30) &&
                       rounds >= 1,
                   SEMIPD_EINVAL, "top_k_top_p_sampling_from_probs: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(probs, vocab))
    hipLaunchKernelGGL((top_k_top_p_sample_kernel<true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream),
                       probs, uniform_samples, top_ks, top_k_val, top_ps, top_p_val, out_ids, success, (int)batch,
                       (int)vocab, rounds);
  else
    hipLaunchKernelGGL((top_k_top_p_sample_kernel<false>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream),
                       probs, uniform_samples, top_ks, top_k_val, top_ps, top_p_val, out_ids, success, (int)batch,
                       (int)vocab, rounds);
  return launch_status("top_k_top_p_sampling_from_probs");
}

int semipd_min_p_sampling_from_probs(const float* probs, const float* uniform_samples, const float* min_ps,
                                     float min_p_val, int32_t* out_ids, int64_t batch, int64_t vocab,
                                     void* stream) {
  SEMIPD_CHECK_ARG(probs && uniform_samples && out_ids && batch >= 0 && vocab > 0 && vocab < (1ll << 30),
                   SEMIPD_EINVAL, "min_p_sampling_from_probs: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(probs, vocab))
    hipLaunchKernelGGL((min_p_sample_kernel<true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), probs,
                       uniform_samples, min_ps, min_p_val, out_ids, (int)vocab);
  else
    hipLaunchKernelGGL((min_p_sample_kernel<false>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), probs,
                       uniform_samples, min_ps, min_p_val, out_ids, (int)vocab);
  return launch_status("min_p_sampling_from_probs");
}

int semipd_top_k_renorm_prob(const float* probs, float* out, const int32_t* top_ks, int32_t top_k_val,
                             int64_t batch, int64_t vocab, void* stream) {
  SEMIPD_CHECK_ARG(probs && out && batch >= 0 && vocab > 0 && vocab < (1ll << 24), SEMIPD_EINVAL,
                   "top_k_renorm_prob: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(probs, vocab) && aligned16(out))
    hipLaunchKernelGGL((renorm_kernel<true, false>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), probs,
                       out, top_ks, top_k_val, (const float*)nullptr, 0.f, (int)vocab);
  else
    hipLaunchKernelGGL((renorm_kernel<false, false>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream),
                       probs, out, top_ks, top_k_val, (const float*)nullptr, 0.f, (int)vocab);
  return launch_status("top_k_renorm_prob");
}

int semipd_token_logprobs(const float* logits, const int32_t* ids, float* out, float* lse, int64_t batch,
                          int64_t vocab, void* stream) {
  SEMIPD_CHECK_ARG(logits && ids && out && batch >= 0 && vocab > 0 && vocab < (1ll << 30), SEMIPD_EINVAL,
                   "token_logprobs: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(logits, vocab))
    hipLaunchKernelGGL((token_logprob_kernel<true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), logits,
                       ids, out, lse, (int)vocab);
  else
    hipLaunchKernelGGL((token_logprob_kernel<false>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), logits,
                       ids, out, lse, (int)vocab);
  return launch_status("token_logprobs");
}

int semipd_top_p_renorm_prob(const float* probs, float* out, const float* top_ps, float top_p_val,
                             int64_t batch, int64_t vocab, void* stream) {
  SEMIPD_CHECK_ARG(probs && out && batch >= 0 && vocab > 0 && vocab < (1ll << 30), SEMIPD_EINVAL,
                   "top_p_renorm_prob: bad arguments");
  if (batch == 0) return 0;
  if (rows_vec_ok(probs, vocab) && aligned16(out))
    hipLaunchKernelGGL((renorm_kernel<true, true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), probs,
                       out, (const int32_t*)nullptr, 0, top_ps, top_p_val, (int)vocab);
  else
    hipLaunchKernelGGL((renorm_kernel<false, true>), dim3((unsigned)batch), dim3(kST), 0, as_stream(stream), probs,
                       out, (const int32_t*)nullptr, 0, top_ps, top_p_val, (int)vocab);
  return launch_status("top_p_renorm_prob");
}

}  // extern "C"
