// MoE routing, token alignment and the expert-grouped GEMM for gfx950,
// plus the lm_head GEMM + greedy argmax that reuses the same MFMA tile (a8/a9).
#include "common.h"

#include <cstdlib>

#include <stdlib.h>

namespace semipd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

union Frag16m {
  uint4 u;
  uint16_t e[8];
  bf16x8_t b;
  f16x8_t f;
};
template <typename T> struct MfmaG;
template <> struct MfmaG<bf16_t> {
  __device__ static inline f32x16 mma(const Frag16m& a, const Frag16m& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct MfmaG<f16_t> {
  __device__ static inline f32x16 mma(const Frag16m& a, const Frag16m& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f, b.f, c, 0, 0, 0);
  }
};

// ---------------------------------------------------------------------------
// Routing: one wave per token.  Scores live in LDS (E <= 512).
// ---------------------------------------------------------------------------
constexpr int kMaxExperts = 512;
constexpr int kMaxTopk = 32;

// Wave-wide argmax (largest value, smallest index among equals) with DPP moves instead of ds_bpermute: the routing kernel
// is one wave per token and a chain of ~100 dependent cross-lane steps (8 argmaxes of 12 shuffles each at top-8); a
// ds_bpermute costs an LDS-crossbar round trip, a DPP move a few cycles.  quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror,
// row_mirror make every row of 16 lanes uniform; row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the result to
// lane 63, which every lane reads.  max / argmax are exact, so the order of the combination does not matter.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmax_dpp_step(float& v, int& i) {
  const float ov = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
  const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
  const bool take = ov > v || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}
__device__ inline void wave_argmax(float& v, int& i) {
  argmax_dpp_step<0xB1, 0xf>(v, i);    // quad_perm [1,0,3,2]
  argmax_dpp_step<0x4E, 0xf>(v, i);    // quad_perm [2,3,0,1]
  argmax_dpp_step<0x141, 0xf>(v, i);   // row_half_mirror
  argmax_dpp_step<0x140, 0xf>(v, i);   // row_mirror
  argmax_dpp_step<0x142, 0xa>(v, i);   // row_bcast:15 into rows 1 and 3
  argmax_dpp_step<0x143, 0xc>(v, i);   // row_bcast:31 into rows 2 and 3
  v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  i = __builtin_amdgcn_readlane(i, 63);
}

// mode: 0 = plain softmax top-k (fused_topk); 1 = grouped (grouped_topk / biased_grouped_topk)
// One wave per token; lane l owns experts l, l + 64, ... (up to kMaxExperts / 64 = 8) and keeps their logits, scores and
// selection values in REGISTERS: the routing of a decode batch is a chain of dependent steps, and every LDS round trip in
// it (the first form of this kernel kept everything in LDS: ~60 of them, 16.5 us at 256 experts in 8 groups) costs more
// than the arithmetic.  LDS is left for the group stage only (a group's experts sit in other lanes).
template <typename T>
__global__ void __launch_bounds__(64)
moe_topk_kernel(const T* __restrict__ gating, const float* __restrict__ bias,
                float* __restrict__ topk_weights, int32_t* __restrict__ topk_ids, int E, int topk,
                int num_group, int topk_group, int renormalize, int scoring, int grouped,
                const float* __restrict__ planes = nullptr, int n_planes = 0, int64_t plane_elems = 0) {
  constexpr int NE = kMaxExperts / 64;
  __shared__ float choice[kMaxExperts];  // group stage: the selection values of every expert
  __shared__ float gscore[64];
  __shared__ int gsel[64];
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x;
  float logit[NE], score[NE], ch[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) logit[i] = 0.f;
  // the router logits of this token as T values: from the [tokens, E] tensor, or -- planes != nullptr -- from the fp32 K-slice
  // planes [n_planes][tokens][E] of the router GEMM (csrc/stream_linear.hip), summed in slice order and rounded to T: the
  // bits that GEMM's own reduction would have written
  if (planes) {
    // 4 experts x 8 planes of loads in flight per lane, added in slice order.  (A plain loop over the planes waits for every
    // load before it issues the next: 4 experts x 14 planes of DeepSeek-V3's router were 56 round trips, 21 us.)
#pragma unroll
    for (int i0 = 0; i0 < NE; i0 += 4) {
      if (i0 * 64 >= E) break;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = 0; z0 < n_planes; z0 += 8) {
        float v[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = lane + (i0 + i) * 64;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            v[i][j] = (e < E && z0 + j < n_planes) ? planes[(int64_t)(z0 + j) * plane_elems + t * E + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (z0 + j == 0) acc[i] = v[i][j];            // (the first plane is the start value, as in splitk_planes_reduce)
            else if (z0 + j < n_planes) acc[i] += v[i][j];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) logit[i0 + i] = Elem<T>::to_f(Elem<T>::from_f(acc[i]));
    }
  } else {
    const T* g = gating + t * E;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = lane + i * 64;
      logit[i] = e < E ? Elem<T>::to_f(g[e]) : 0.f;
    }
  }
  // scores (unbiased: the weights come from these)
  if (scoring == 0) {
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if (lane + i * 64 < E) mx = fmaxf(mx, logit[i]);
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      score[i] = 0.f;
      if (lane + i * 64 < E) {
        score[i] = expf(logit[i] - mx);
        sum += score[i];
      }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < NE; ++i) score[i] *= inv;
  } else {
#pragma unroll
    for (int i = 0; i < NE; ++i) score[i] = 1.f / (1.f + expf(-logit[i]));
  }
  // grouped_topk / biased_grouped_topk run softmax / sigmoid in the gating dtype (topk.py:91-94,
  // 132): scores are rounded to T before any comparison so the selection matches bit for bit.
  if (grouped) {
#pragma unroll
    for (int i = 0; i < NE; ++i) score[i] = Elem<T>::to_f(Elem<T>::from_f(score[i]));
  }
  // selection values (biased / masked)
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = lane + i * 64;
    ch[i] = e < E ? score[i] + (bias ? bias[e] : 0.f) : -INFINITY;
  }
  // (one group: it is the group that is selected and nothing is masked -- DeepSeek-V2-Lite)
  if (grouped && num_group > 1) {
    const int gs = E / num_group;
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if (lane + i * 64 < E) choice[lane + i * 64] = ch[i];
    __syncthreads();
    float mine = -INFINITY;
    if (lane < num_group) {
      float best = -INFINITY, second = -INFINITY;
      for (int j = 0; j < gs; ++j) {   // branch-free top two: the loads pipeline
        const float x = choice[lane * gs + j];
        second = fmaxf(second, fminf(best, x));
        best = fmaxf(best, x);
      }
      mine = bias ? best + second : best;  // topk.py:140-144 vs :98-100
      gscore[lane] = mine;
    }
    __syncthreads();
    // the topk_group best groups, ties to the lower index (torch.topk on the group scores as the serial argmax chain took
    // them): group g is selected when fewer than topk_group groups come before it in that order -- every lane ranks its own
    if (lane < num_group) {
      int before = 0;
      for (int j = 0; j < num_group; ++j) {
        const float o = gscore[j];
        before += (o > mine || (o == mine && j < lane)) ? 1 : 0;
      }
      gsel[lane] = before < topk_group ? 1 : 0;
    }
    __syncthreads();
    const float fill = bias ? -INFINITY : 0.f;  // masked_fill value (topk.py:151-153 vs :110)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = lane + i * 64;
      if (e < E && !gsel[e / gs]) ch[i] = fill;
    }
  }
  // iterative top-k over the selection values
  float wsum = 0.f;
  float my_w = 0.f;
  int my_id = 0;
  for (int k = 0; k < topk; ++k) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = lane + i * 64;
      const float x = ch[i];
      if (e < E && (x > bv || (x == bv && e < bi))) {
        bv = x;
        bi = e;
      }
    }
    wave_argmax(bv, bi);
    if (bi == 0x7fffffff) bi = 0;
    const int owner = bi & 63, slot = bi >> 6;   // (uniform: wave_argmax ends in a readlane)
    float w = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      if (i == slot) {
        w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(score[i]), owner));
        if (lane == owner) ch[i] = -INFINITY;
      }
    }
    wsum += w;
    if (lane == k) {
      my_w = w;
      my_id = bi;
    }
  }
  if (lane < topk) {
    float wgt = my_w;
    if (renormalize) {
      if (grouped) {  // division happens in the gating dtype, then .to(float32) (topk.py:114-117)
        const float s = Elem<T>::to_f(Elem<T>::from_f(wsum));
        wgt = Elem<T>::to_f(Elem<T>::from_f(my_w / s));
      } else {
        wgt = my_w / wsum;
      }
    }
    topk_weights[t * topk + lane] = wgt;
    topk_ids[t * topk + lane] = my_id;
  }
}

// ---------------------------------------------------------------------------
// moe_align_block_size
// ---------------------------------------------------------------------------
__global__ void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}

// single workgroup: histogram + padded exclusive scan + expert_ids; optionally (FUSED) also the
// sentinel fill and the scatter, for decode-sized inputs.
template <bool FUSED>
__global__ void __launch_bounds__(1024)
moe_align_kernel(const int32_t* __restrict__ topk_ids, int64_t numel, int E, int block_size,
                 int32_t* __restrict__ sorted_ids, int32_t* __restrict__ expert_ids,
                 int32_t* __restrict__ num_post_pad, int32_t* __restrict__ cumsum,
                 int64_t max_sorted) {
  __shared__ int hist[kMaxExperts];
  __shared__ int offs[kMaxExperts + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < E; e += blockDim.x) hist[e] = 0;
  if (FUSED)
    for (int64_t i = tid; i < max_sorted; i += blockDim.x) sorted_ids[i] = (int32_t)numel;
  __syncthreads();
  for (int64_t i = tid; i < numel; i += blockDim.x) {
    const int e = topk_ids[i];
    if (e >= 0 && e < E) atomicAdd(&hist[e], 1);
  }
  __syncthreads();
  // padded exclusive scan over the experts by the first wave: lane l owns `per` consecutive experts, the lane totals are
  // scanned with shuffles (one thread walking 256 experts with a division each was 9 of the kernel's 15 us at E = 256)
  if (tid < 64) {
    const int per = (E + 63) / 64;
    const int e0 = tid * per;
    int local = 0;
    for (int j = 0; j < per; ++j) {
      const int e = e0 + j;
      if (e < E) local += (hist[e] + block_size - 1) / block_size * block_size;
    }
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (tid >= o) incl += up;
    }
    int run = incl - local;
    for (int j = 0; j < per; ++j) {
      const int e = e0 + j;
      if (e < E) {
        offs[e] = run;
        run += (hist[e] + block_size - 1) / block_size * block_size;
      }
    }
    if (tid == 63) {
      offs[E] = incl;
      *num_post_pad = incl;
    }
  }
  __syncthreads();
  for (int e = tid; e <= E; e += blockDim.x) cumsum[e] = offs[e];
  // expert id of every block
  const int nblocks = offs[E] / block_size;
  for (int e = tid; e < E; e += blockDim.x)
    for (int b = offs[e] / block_size; b < offs[e + 1] / block_size; ++b) expert_ids[b] = e;
  (void)nblocks;
  if (FUSED) {
    __syncthreads();
    for (int e = tid; e < E; e += blockDim.x) hist[e] = offs[e];  // running write cursors
    __syncthreads();
    for (int64_t i = tid; i < numel; i += blockDim.x) {
      const int e = topk_ids[i];
      if (e >= 0 && e < E) {
        const int pos = atomicAdd(&hist[e], 1);
        sorted_ids[pos] = (int32_t)i;
      }
    }
  }
}

// Every workgroup takes a contiguous chunk of kScatterPerThread ids per thread, ranks them per expert in LDS and reserves
// ONE range per (workgroup, expert) in the global cursors: one global atomic per 2048 ids and expert instead of one per
// id (49 k atomics on 64 addresses took 60 us of a T = 8192 call).  The order inside an expert's rows stays undefined,
// as in the reference (moe_align_kernel.cu:77-95).
constexpr int kScatterPerThread = 8;

__global__ void __launch_bounds__(256)
moe_scatter_kernel(const int32_t* __restrict__ topk_ids, int64_t numel, int E, int32_t* __restrict__ sorted_ids,
                   int32_t* __restrict__ cursor) {
  __shared__ int cnt[kMaxExperts];
  const int tid = threadIdx.x;
  for (int e = tid; e < E; e += 256) cnt[e] = 0;
  __syncthreads();
  const int64_t base_i = (int64_t)blockIdx.x * (256 * kScatterPerThread);
  int ex[kScatterPerThread], rank[kScatterPerThread];
#pragma unroll
  for (int j = 0; j < kScatterPerThread; ++j) {
    const int64_t i = base_i + j * 256 + tid;
    ex[j] = -1;
    if (i < numel) {
      const int e = topk_ids[i];
      if (e >= 0 && e < E) {
        ex[j] = e;
        rank[j] = atomicAdd(&cnt[e], 1);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    const int n = cnt[e];
    cnt[e] = n ? atomicAdd(&cursor[e], n) : 0;    // now the base of this workgroup's range
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kScatterPerThread; ++j)
    if (ex[j] >= 0) sorted_ids[cnt[ex[j]] + rank[j]] = (int32_t)(base_i + j * 256 + tid);
}

// ---------------------------------------------------------------------------
// C[row, n] = sum_k A[arow, k] * W[expert, n, k]      ("NT" GEMM, W row-major [N,K])
// Workgroup tile 64 x 256, BK = 64, 4 waves each 64 x 64 (2 x 2 tiles of v_mfma_f32_32x32x16).
// The next K-tile (A rows gathered through the sorted token ids, W rows of the block's expert) is
// fetched into registers while the current one is multiplied out of LDS (issue early / write late),
// so a decode-sized call streams the expert weights once at HBM rate and a prefill-sized call keeps
// the matrix pipes fed.  GROUPED: rows come from sorted_token_ids / expert_ids (fused_moe_kernel
// semantics, fused_moe.py:54-273); otherwise plain dense rows (lm_head).
// ---------------------------------------------------------------------------
template <typename T, typename OutT, bool GROUPED, int BN>
__global__ void __launch_bounds__(256)
gemm_nt_kernel(OutT* __restrict__ c, const T* __restrict__ a, const T* __restrict__ w,
               const float* __restrict__ topk_weights, const int32_t* __restrict__ sorted_ids,
               const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
               int64_t num_valid, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc,
               int top_k_div, int mul_routed_weight) {
  constexpr int BM = 64, BK = 64;
  constexpr int NJ = BN / 128;              // 32-column tiles per wave (wave owns BN/4 columns)
  constexpr int AS = BK + 8, WS = BK + 8;  // row strides (elements), 16 B pad: conflict-free ds_read_b128
  constexpr int NA = BM * (BK / 8) / 256;  // 2 A chunks per thread per K-tile
  constexpr int NW = BN * (BK / 8) / 256;  // 8 W chunks per thread per K-tile
  __shared__ __attribute__((aligned(16))) uint16_t a_lds[BM * AS];
  __shared__ __attribute__((aligned(16))) uint16_t w_lds[BN * WS];
  __shared__ int row_id[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, hi = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int64_t n0 = (int64_t)blockIdx.x * BN;
  int64_t expert = 0;
  if (GROUPED) {
    if (m0 >= *num_post_pad) return;
    expert = expert_ids[blockIdx.y];
    if (tid < BM) {
      const int sid = sorted_ids[m0 + tid];
      row_id[tid] = (sid >= 0 && sid < num_valid) ? sid : -1;
    }
  } else {
    if (tid < BM) row_id[tid] = (m0 + tid < M) ? (int)(m0 + tid) : -1;
  }
  __syncthreads();
  const T* wbase = w + expert * N * K;

  // this thread's staging slots: chunk ch = tid & 7 (8 elements) of rows (tid >> 3) + 32*i
  const int ch = tid & 7, r0 = tid >> 3;
  const T* a_ptr[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int rid = row_id[r0 + 32 * i];
    a_ok[i] = rid >= 0;
    const int64_t arow = GROUPED ? (int64_t)(rid / top_k_div) : (int64_t)rid;
    a_ptr[i] = a + (a_ok[i] ? arow : 0) * lda + ch * 8;
  }
  const T* w_ptr[NW];
  bool w_ok[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int64_t n = n0 + r0 + 32 * i;
    w_ok[i] = n < N;
    w_ptr[i] = wbase + (w_ok[i] ? n : 0) * K + ch * 8;
  }
  const bool k_vec = (K % 8 == 0) && (lda % 8 == 0);

  Frag16m areg[NA], wreg[NW];
  auto fetch = [&](int64_t k0) __attribute__((always_inline)) {
    const int64_t kk = k0 + ch * 8;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      areg[i].u = make_uint4(0, 0, 0, 0);
      if (a_ok[i]) {
        if (k_vec && kk + 8 <= K) {
          areg[i].u = *reinterpret_cast<const uint4*>(a_ptr[i] + k0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) areg[i].e[j] = (kk + j < K) ? a_ptr[i][k0 + j].v : (uint16_t)0;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      wreg[i].u = make_uint4(0, 0, 0, 0);
      if (w_ok[i]) {
        if (k_vec && kk + 8 <= K) {
          wreg[i].u = *reinterpret_cast<const uint4*>(w_ptr[i] + k0);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) wreg[i].e[j] = (kk + j < K) ? w_ptr[i][k0 + j].v : (uint16_t)0;
        }
      }
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const uint16_t* a_rd = &a_lds[col * AS + hi * 8];
  const uint16_t* w_rd = &w_lds[(wave * (BN / 4) + col) * WS + hi * 8];
  fetch(0);
  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // previous tile consumed
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<uint4*>(&a_lds[(r0 + 32 * i) * AS + ch * 8]) = areg[i].u;
#pragma unroll
    for (int i = 0; i < NW; ++i)
      *reinterpret_cast<uint4*>(&w_lds[(r0 + 32 * i) * WS + ch * 8]) = wreg[i].u;
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);  // in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      Frag16m a0, a1;
      a0.u = *reinterpret_cast<const uint4*>(a_rd + ks * 16);
      a1.u = *reinterpret_cast<const uint4*>(a_rd + 32 * AS + ks * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        Frag16m b;
        b.u = *reinterpret_cast<const uint4*>(w_rd + j * 32 * WS + ks * 16);
        acc[0][j] = MfmaG<T>::mma(a0, b, acc[0][j]);
        acc[1][j] = MfmaG<T>::mma(a1, b, acc[1][j]);
      }
    }
  }
  // epilogue: lane holds C[m = i*32 + (r&3)+8*(r>>2)+4*hi][n = n0 + wave*(BN/4) + j*32 + col]
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int64_t n = n0 + wave * (BN / 4) + j * 32 + col;
    if (n >= N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int rid = row_id[m];
        if (rid >= 0) {
          float v = acc[i][j][r];
          if (GROUPED && mul_routed_weight) v *= topk_weights[rid];
          OutT* dst = c + (int64_t)rid * ldc + n;
          if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float*>(dst) = v;
          else *dst = Elem<OutT>::from_f(v);
        }
      }
    }
  }
}

// defined in skinny_gemm.hip
bool skinny_gemm_ok(int64_t K, int64_t lda, const void* a, const void* w);
template <typename T, typename OutT, bool GROUPED, int BM = 64, int NG = (BM == 128 ? 4 : 1)>
int launch_skinny_gemm(OutT* c, const T* a, const T* w, const float* topk_weights, const int32_t* sorted_ids,
                       const int32_t* expert_ids, const int32_t* num_post_pad, int64_t num_valid, int64_t M,
                       int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t m_blocks, int top_k_div,
                       int mul_routed_weight, hipStream_t st, int ksplit, float* partial_ws);
int skinny_pick_ksplit(int64_t rows, int64_t N, int64_t K, int64_t m_blocks, int num_cus, int rows_per_wg);

}  // namespace semipd

namespace semipd {
template <typename T>
int launch_moe_tiled_gemm(T* c, const T* a, const T* w, const float* topk_weights, const int32_t* sorted_ids,
                          const int32_t* expert_ids, const int32_t* num_post_pad, int64_t num_valid, int64_t n, int64_t k,
                          int64_t max_sorted, int top_k_div, int mul_routed_weight, hipStream_t st);
bool moe_tiled_gemm_silu_supported(int64_t num_valid, int64_t n, int64_t k, int top_k_div);
template <typename T>
int launch_moe_tiled_gemm_silu(T* c, const T* a, const T* w, const int32_t* sorted_ids, const int32_t* expert_ids,
                               const int32_t* num_post_pad, int64_t num_valid, int64_t n, int64_t k, int64_t max_sorted,
                               int top_k_div, hipStream_t st);
}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_topk_softmax(const void* gating, float* topk_weights, int32_t* topk_ids,
                        int64_t num_tokens, int num_experts, int topk, int renormalize, int dtype,
                        void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && num_experts > 0 && num_experts <= kMaxExperts && topk > 0 &&
                       topk <= kMaxTopk && topk <= num_experts,
                   SEMIPD_EINVAL, "topk_softmax: bad sizes (E<=%d, topk<=%d)", kMaxExperts, kMaxTopk);
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(gating && topk_weights && topk_ids, SEMIPD_EINVAL, "topk_softmax: null pointer");
  SEMIPD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((moe_topk_kernel<T>), dim3((unsigned)num_tokens), dim3(64), 0, as_stream(stream), (const T*)gating, (const float*)nullptr, topk_weights, topk_ids, num_experts, topk, 1, 1, renormalize, 0, 0));
  return launch_status("topk_softmax");
}

int semipd_grouped_topk(const void* gating, const float* correction_bias, float* topk_weights,
                        int32_t* topk_ids, int64_t num_tokens, int num_experts, int topk,
                        int num_expert_group, int topk_group, int renormalize, int scoring,
                        int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && num_experts > 0 && num_experts <= kMaxExperts && topk > 0 &&
                       topk <= kMaxTopk && topk <= num_experts && num_expert_group > 0 &&
                       num_expert_group <= 64 && num_experts % num_expert_group == 0 &&
                       topk_group > 0 && topk_group <= num_expert_group,
                   SEMIPD_EINVAL, "grouped_topk: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(gating && topk_weights && topk_ids, SEMIPD_EINVAL, "grouped_topk: null pointer");
  const int sc = correction_bias ? 1 : scoring;  // biased_grouped_topk always uses sigmoid
  SEMIPD_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((moe_topk_kernel<T>), dim3((unsigned)num_tokens), dim3(64), 0, as_stream(stream), (const T*)gating, correction_bias, topk_weights, topk_ids, num_experts, topk, num_expert_group, topk_group, renormalize, sc, 1));
  return launch_status("grouped_topk");
}

int semipd_grouped_topk_planes(const float* planes, int n_planes, int64_t plane_elems, const float* correction_bias,
                               float* topk_weights, int32_t* topk_ids, int64_t num_tokens, int num_experts, int topk,
                               int num_expert_group, int topk_group, int renormalize, int scoring, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_tokens >= 0 && num_experts > 0 && num_experts <= kMaxExperts && topk > 0 &&
                       topk <= kMaxTopk && topk <= num_experts && num_expert_group > 0 &&
                       num_expert_group <= 64 && num_experts % num_expert_group == 0 &&
                       topk_group > 0 && topk_group <= num_expert_group && n_planes >= 1 && n_planes <= 64 &&
                       plane_elems >= num_tokens * num_experts,
                   SEMIPD_EINVAL, "grouped_topk_planes: bad sizes");
  if (num_tokens == 0) return 0;
  SEMIPD_CHECK_ARG(planes && topk_weights && topk_ids, SEMIPD_EINVAL, "grouped_topk_planes: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "grouped_topk_planes: the router GEMM's "
                   "dtype (bf16 / f16) expected");
  const int sc = correction_bias ? 1 : scoring;
  SEMIPD_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((moe_topk_kernel<T>), dim3((unsigned)num_tokens), dim3(64), 0, as_stream(stream), (const T*)nullptr, correction_bias, topk_weights, topk_ids, num_experts, topk, num_expert_group, topk_group, renormalize, sc, 1, planes, n_planes, plane_elems));
  return launch_status("grouped_topk_planes");
}

int semipd_moe_align_block_size(const int32_t* topk_ids, int64_t numel, int num_experts,
                                int block_size, int32_t* sorted_token_ids, int32_t* expert_ids,
                                int32_t* num_tokens_post_pad, int32_t* cumsum_buffer,
                                int64_t max_sorted, void* stream) {
  SEMIPD_CHECK_ARG(numel >= 0 && num_experts > 0 && num_experts <= kMaxExperts && block_size > 0,
                   SEMIPD_EINVAL, "moe_align_block_size: bad sizes");
  SEMIPD_CHECK_ARG(topk_ids && sorted_token_ids && expert_ids && num_tokens_post_pad && cumsum_buffer,
                   SEMIPD_EINVAL, "moe_align_block_size: null pointer");
  SEMIPD_CHECK_ARG(max_sorted >= numel + (int64_t)num_experts * (block_size - 1), SEMIPD_EINVAL, "moe_align_block_size: sorted_token_ids too small");
  hipStream_t st = as_stream(stream);
  if (numel <= 4096 && max_sorted <= 65536) {
    hipLaunchKernelGGL((moe_align_kernel<true>), dim3(1), dim3(1024), 0, st, topk_ids, numel,
                       num_experts, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad,
                       cumsum_buffer, max_sorted);
    return launch_status("moe_align(fused)");
  }
  int fb = (int)((max_sorted + 255) / 256);
  if (fb > 2048) fb = 2048;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(fb), dim3(256), 0, st, sorted_token_ids, max_sorted,
                     (int32_t)numel);
  hipLaunchKernelGGL((moe_align_kernel<false>), dim3(1), dim3(1024), 0, st, topk_ids, numel,
                     num_experts, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad,
                     cumsum_buffer, max_sorted);
  const int sb = (int)((numel + 256 * kScatterPerThread - 1) / (256 * kScatterPerThread));
  // cumsum_buffer[0..E) doubles as the running write cursor (as in moe_align_kernel.cu:77-95);
  // after the call it holds the *end* offsets of each expert's real tokens.
  hipLaunchKernelGGL(moe_scatter_kernel, dim3(sb), dim3(256), 0, st, topk_ids, numel, num_experts,
                     sorted_token_ids, cumsum_buffer);
  return launch_status("moe_align");
}

int semipd_moe_grouped_gemm(void* c, const void* a, const void* w, const float* topk_weights,
                            const int32_t* sorted_token_ids, const int32_t* expert_ids,
                            const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                            int64_t k, int64_t max_sorted, int top_k_div, int mul_routed_weight,
                            int block_m, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(n > 0 && k > 0 && num_valid >= 0 && max_sorted >= 0 && top_k_div > 0,
                   SEMIPD_EINVAL, "moe_grouped_gemm: bad sizes");
  SEMIPD_CHECK_ARG(block_m == 64 || block_m == 128, SEMIPD_ESHAPE, "moe_grouped_gemm: block_m must be 64 or 128");
  if (num_valid == 0 || max_sorted == 0) return 0;
  SEMIPD_CHECK_ARG(c && a && w && sorted_token_ids && expert_ids && num_tokens_post_pad, SEMIPD_EINVAL,
                   "moe_grouped_gemm: null pointer");
  SEMIPD_CHECK_ARG(!mul_routed_weight || topk_weights, SEMIPD_EINVAL,
                   "moe_grouped_gemm: topk_weights required");
  SEMIPD_CHECK_ARG(aligned16(a) && aligned16(w), SEMIPD_EALIGN, "moe_grouped_gemm: unaligned pointer");
  // Up to a few hundred rows per expert the call is bound by streaming the expert weights (one pass per
  // block of block_m rows), not by MFMA: weight-streaming kernel, 64-row blocks for decode, 128-row
  // blocks for prefill chunks
  if (skinny_gemm_ok(k, k, a, w) && n % 4 == 0) {
    // prefill-sized calls (hundreds of rows per expert): the tiled kernel (moe_tiled_gemm.hip); SEMIPD_MOE_TILED=0 keeps
    // the streaming kernel for every size
    static const bool tiled_on = [] { const char* e = getenv("SEMIPD_MOE_TILED"); return !(e && e[0] == '0'); }();
    if (tiled_on && block_m == 128 && num_valid >= 2048 && (reinterpret_cast<uintptr_t>(c) & 7u) == 0) {
      int miss = 1;
      SEMIPD_DISPATCH_HALF(dtype, T, miss = (launch_moe_tiled_gemm<T>((T*)c, (const T*)a, (const T*)w, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, n, k, max_sorted, top_k_div, mul_routed_weight, as_stream(stream))));
      if (!miss) return launch_status("moe_grouped_gemm(tiled)");
    }
    if (block_m == 128) {
      SEMIPD_DISPATCH_HALF(dtype, T, return (launch_skinny_gemm<T, T, true, 128>((T*)c, (const T*)a, (const T*)w, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, (int64_t)0, n, k, k, n, (max_sorted + 127) / 128, top_k_div, mul_routed_weight, as_stream(stream), 1, nullptr)));
    }
    SEMIPD_DISPATCH_HALF(dtype, T, return (launch_skinny_gemm<T, T, true, 64>((T*)c, (const T*)a, (const T*)w, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, (int64_t)0, n, k, k, n, (max_sorted + 63) / 64, top_k_div, mul_routed_weight, as_stream(stream), 1, nullptr)));
  }
  SEMIPD_CHECK_ARG(block_m == 64, SEMIPD_ESHAPE,
                   "moe_grouped_gemm: block_m 128 needs k %% 32 == 0, n %% 4 == 0 and 16-byte aligned rows");
  // general tiled kernel for the shapes the streaming kernel does not take
  const bool wide = num_valid >= 2048;
  dim3 grid((unsigned)((n + (wide ? 255 : 127)) / (wide ? 256 : 128)), (unsigned)((max_sorted + 63) / 64));
  if (wide) {
    SEMIPD_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((gemm_nt_kernel<T, T, true, 256>), grid, dim3(256), 0, as_stream(stream), (T*)c, (const T*)a, (const T*)w, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, (int64_t)0, n, k, k, n, top_k_div, mul_routed_weight));
  } else {
    SEMIPD_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((gemm_nt_kernel<T, T, true, 128>), grid, dim3(256), 0, as_stream(stream), (T*)c, (const T*)a, (const T*)w, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, (int64_t)0, n, k, k, n, top_k_div, mul_routed_weight));
  }
  return launch_status("moe_grouped_gemm");
}

int semipd_moe_grouped_gemm_silu_supported(int64_t num_valid, int64_t n, int64_t k, int top_k_div, int block_m, int dtype) {
  static const bool tiled_on = [] { const char* e = getenv("SEMIPD_MOE_TILED"); return !(e && e[0] == '0'); }();
  return tiled_on && block_m == 128 && num_valid >= 2048 && (dtype == SEMIPD_BF16 || dtype == SEMIPD_F16) &&
         moe_tiled_gemm_silu_supported(num_valid, n, k, top_k_div) ? 1 : 0;
}

int semipd_moe_grouped_gemm_silu(void* c, const void* a, const void* w, const int32_t* sorted_token_ids,
                                 const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                                 int64_t k, int64_t max_sorted, int top_k_div, int block_m, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(n > 0 && k > 0 && num_valid >= 0 && max_sorted >= 0 && top_k_div > 0, SEMIPD_EINVAL,
                   "moe_grouped_gemm_silu: bad sizes");
  if (num_valid == 0 || max_sorted == 0) return 0;
  SEMIPD_CHECK_ARG(c && a && w && sorted_token_ids && expert_ids && num_tokens_post_pad, SEMIPD_EINVAL,
                   "moe_grouped_gemm_silu: null pointer");
  SEMIPD_CHECK_ARG(semipd_moe_grouped_gemm_silu_supported(num_valid, n, k, top_k_div, block_m, dtype), SEMIPD_ESHAPE,
                   "moe_grouped_gemm_silu: prefill-sized calls only (block_m 128, >= 2048 routed rows, k %% 64, n %% 64); "
                   "use semipd_moe_grouped_gemm + semipd_silu_and_mul");
  SEMIPD_CHECK_ARG(aligned16(a) && aligned16(w) && (reinterpret_cast<uintptr_t>(c) & 7u) == 0, SEMIPD_EALIGN,
                   "moe_grouped_gemm_silu: unaligned pointer");
  int miss = 1;
  SEMIPD_DISPATCH_HALF(dtype, T, miss = (launch_moe_tiled_gemm_silu<T>((T*)c, (const T*)a, (const T*)w, sorted_token_ids, expert_ids, num_tokens_post_pad, num_valid, n, k, max_sorted, top_k_div, as_stream(stream))));
  SEMIPD_CHECK_ARG(!miss, SEMIPD_ESHAPE, "moe_grouped_gemm_silu: shape not covered");
  return launch_status("moe_grouped_gemm_silu");
}

size_t semipd_lm_head_argmax_workspace(int64_t batch, int64_t vocab) {
  return (size_t)batch * (size_t)vocab * sizeof(float);
}

int semipd_argmax(const void* logits, void* out, int64_t batch, int64_t vocab,
                  int64_t logits_stride, int dtype, int out_is_i64, void* stream);

int semipd_lm_head_argmax(const void* hidden, const void* weight, float* logits, void* out,
                          void* workspace, int64_t batch, int64_t hidden_size, int64_t vocab,
                          int dtype, int out_is_i64, void* stream) {
  SEMIPD_CHECK_ARG(batch >= 0 && hidden_size > 0 && vocab > 0, SEMIPD_EINVAL,
                   "lm_head_argmax: bad sizes");
  if (batch == 0) return 0;
  SEMIPD_CHECK_ARG(hidden && weight && out, SEMIPD_EINVAL, "lm_head_argmax: null pointer");
  float* lg = logits ? logits : (float*)workspace;
  SEMIPD_CHECK_ARG(lg, SEMIPD_EINVAL, "lm_head_argmax: logits or workspace required");
  SEMIPD_CHECK_ARG(aligned16(hidden) && aligned16(weight), SEMIPD_EALIGN,
                   "lm_head_argmax: unaligned pointer");
  if (batch <= 64 && hidden_size % 128 == 0 && vocab % 16 == 0 && aligned16(lg)) {
    // the LDS-DMA streaming kernel (csrc/stream_linear.hip) with its fp32 accumulators stored as they are: ~1.6x the
    // per-CU weight rate of the register-fragment kernel below on a partial CU share
    int rc0 = semipd_stream_linear_f32(lg, hidden, weight, batch, vocab, hidden_size, hidden_size, dtype, stream);
    if (rc0) return rc0;
    return semipd_argmax(lg, out, batch, vocab, vocab, SEMIPD_F32, out_is_i64, stream);
  }
  if (batch <= 64 && skinny_gemm_ok(hidden_size, hidden_size, hidden, weight) && vocab % 4 == 0) {
    int rc0 = 0;
    SEMIPD_DISPATCH_HALF(dtype, T, rc0 = (launch_skinny_gemm<T, float, false>(lg, (const T*)hidden, (const T*)weight, nullptr, nullptr, nullptr, nullptr, (int64_t)0, batch, vocab, hidden_size, hidden_size, vocab, (int64_t)1, 1, 0, as_stream(stream), 1, nullptr)));
    if (rc0) return rc0;
    return semipd_argmax(lg, out, batch, vocab, vocab, SEMIPD_F32, out_is_i64, stream);
  }
  dim3 grid((unsigned)((vocab + 127) / 128), (unsigned)((batch + 63) / 64));
  SEMIPD_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((gemm_nt_kernel<T, float, false, 128>), grid, dim3(256), 0, as_stream(stream), lg, (const T*)hidden, (const T*)weight, (const float*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, (int64_t)0, batch, vocab, hidden_size, hidden_size, vocab, 1, 0));
  int rc = launch_status("lm_head_gemm");
  if (rc) return rc;
  return semipd_argmax(lg, out, batch, vocab, vocab, SEMIPD_F32, out_is_i64, stream);
}

// ---- dense linear at decode batch sizes ---------------------------------------------------------
size_t semipd_linear_workspace(int64_t max_rows, int64_t max_n) {
  // 16 split-K partial planes of [rows rounded to 64, n] fp32
  const int64_t rows = (max_rows + 63) / 64 * 64;
  return (size_t)(16 * rows * max_n * 4);
}

int semipd_linear(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes,
                  int64_t rows, int64_t n, int64_t k, int64_t ldx, int64_t ldo, int num_cus, int dtype,
                  void* stream) {
  SEMIPD_CHECK_ARG(rows >= 0 && n > 0 && k > 0 && ldx >= k && ldo >= n, SEMIPD_EINVAL, "linear: bad sizes");
  if (rows == 0) return 0;
  SEMIPD_CHECK_ARG(out && x && weight, SEMIPD_EINVAL, "linear: null pointer");
  SEMIPD_CHECK_ARG(rows <= 256, SEMIPD_ESHAPE,
                   "linear: %lld rows; this entry point is the weight-streaming path for decode batches (<= 256)",
                   (long long)rows);
  SEMIPD_CHECK_ARG(skinny_gemm_ok(k, ldx, x, weight) && n % 4 == 0 && ldo % 4 == 0 && aligned16(out), SEMIPD_EALIGN,
                   "linear: k %% 32, n %% 4, 16-byte aligned rows required");
  const int64_t m_blocks = (rows + 63) / 64;
  // W rows per workgroup (64 x NG): wide workgroups for wide layers keep the launch within ONE round of
  // workgroup slots on the CUs the process owns (no tail), narrow ones + split-K for the square layers
  int ng = n >= 16384 ? 2 : 1;
  if (const char* e = getenv("SEMIPD_LINEAR_NG")) ng = atoi(e) == 4 ? 4 : atoi(e) == 2 ? 2 : 1;
  int ksplit = 1;
  if (workspace && aligned16(workspace)) {
    ksplit = skinny_pick_ksplit(rows, n, k, m_blocks, num_cus, 64 * ng);
    if (const char* e = getenv("SEMIPD_LINEAR_KSPLIT")) ksplit = atoi(e) > 0 ? atoi(e) : ksplit;
    const size_t plane = (size_t)(m_blocks * 64) * n * 4;
    while (ksplit > 1 && (size_t)ksplit * plane > workspace_bytes) --ksplit;
  }
  float* ws = ksplit > 1 ? (float*)workspace : nullptr;
  if (ng == 4) {
    SEMIPD_DISPATCH_HALF(dtype, T, return (launch_skinny_gemm<T, T, false, 64, 4>((T*)out, (const T*)x, (const T*)weight, nullptr, nullptr, nullptr, nullptr, (int64_t)0, rows, n, k, ldx, ldo, m_blocks, 1, 0, as_stream(stream), ksplit, ws)));
  } else if (ng == 2) {
    SEMIPD_DISPATCH_HALF(dtype, T, return (launch_skinny_gemm<T, T, false, 64, 2>((T*)out, (const T*)x, (const T*)weight, nullptr, nullptr, nullptr, nullptr, (int64_t)0, rows, n, k, ldx, ldo, m_blocks, 1, 0, as_stream(stream), ksplit, ws)));
  }
  SEMIPD_DISPATCH_HALF(dtype, T, return (launch_skinny_gemm<T, T, false, 64, 1>((T*)out, (const T*)x, (const T*)weight, nullptr, nullptr, nullptr, nullptr, (int64_t)0, rows, n, k, ldx, ldo, m_blocks, 1, 0, as_stream(stream), ksplit, ws)));
  return 0;
}

}  // extern "C"
