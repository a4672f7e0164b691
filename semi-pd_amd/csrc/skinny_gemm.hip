// Weight-streaming "skinny" NT GEMM for gfx950: C[m, n] = sum_k A[m, k] * W[e, n, k] with at most 64
// rows of A per block (decode-sized MoE expert blocks, lm_head at decode batch sizes).
//
// These calls are bound by reading W once from HBM, so the kernel is built like the decode-attention
// kernel rather than like a tiled GEMM:
//   * a wave owns 16 rows of W (output columns) and streams them straight from HBM in MFMA A-operand
//     layout (lane l: row l & 15, 16 bytes at k-block l >> 4), one K-chunk of 256 ahead in registers;
//   * the <= 64 activation rows of the block are staged per K-chunk in LDS (shared by the 4 waves,
//     +16 B row pad, conflict-free ds_read_b128) and used as the B operand: C^T[n, m] tiles of
//     v_mfma_f32_16x16x32, up to four m-tiles, empty m-tiles skipped;
//   * a lane ends up with 4 consecutive n of one row m, so the stores are 8 / 16 bytes wide.
// GROUPED = fused_moe_kernel semantics (rows through sorted_token_ids, weights of expert_ids[block],
// optional routed-weight multiply; layers/moe/fused_moe_triton/fused_moe.py:54-273).
#include "common.h"

namespace semipd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

union FragS {
  uint4 u;
  uint16_t e[8];
  bf16x8_t b;
  f16x8_t f;
};
template <typename T> struct MfmaS;
template <> struct MfmaS<bf16_t> {
  __device__ static inline f32x4 mma(const FragS& a, const FragS& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct MfmaS<f16_t> {
  __device__ static inline f32x4 mma(const FragS& a, const FragS& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
  }
};

template <typename T, typename OutT, bool GROUPED, int BM, int NG>
__global__ void __launch_bounds__(256, 2)
skinny_gemm_kernel(OutT* __restrict__ c, const T* __restrict__ a, const T* __restrict__ w,
                   const float* __restrict__ topk_weights, const int32_t* __restrict__ sorted_ids,
                   const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
                   int64_t num_valid, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc,
                   int top_k_div, int mul_routed_weight, int chunks_per_split, float* __restrict__ partial_ws) {
  // BM = 64 (decode blocks, dense calls): a wave owns 16 W rows, K chunks of 256.
  // BM = 128 (prefill chunks of a grouped GEMM): a wave owns FOUR groups of 16 W rows (256 W rows per
  // workgroup) and the K chunk shrinks to 64, so the registers in flight stay the same while every staged
  // activation fragment feeds four MFMAs and every activation row is re-read by a quarter as many
  // workgroups.  With 64 W rows per workgroup the 44x re-read of the activations through L2, not the
  // weights, bounded the prefill call (profiles/r01_kbench_moe_v*.txt: 327 -> 505 TFLOP/s at T = 8192).
  static_assert(NG == 1 || NG == 2 || NG == 4, "NG = groups of 16 W rows per wave");
  constexpr int KC = 256 / NG;                    // K chunk staged per barrier pair
  constexpr int BNW = 16 * NG;                    // W rows per wave
  constexpr int MT = BM / 16;                     // m-tiles of 16 rows
  constexpr int AS = KC + 8;                      // LDS row stride (elements)
  constexpr int KSC = KC / 32;                    // MFMA k-steps per chunk
  constexpr int CPRW = KC / 8;                    // 16-byte chunks per staged row
  constexpr int RPP = 256 / CPRW;                 // rows staged per pass of the 256 threads
  constexpr int NA = BM / RPP;                    // A chunks of 16 B per thread per K-chunk (8)
  __shared__ __attribute__((aligned(16))) uint16_t a_lds[BM * AS];
  __shared__ int row_id[BM];
  __shared__ int n_rows;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int64_t n0 = (int64_t)blockIdx.x * (4 * BNW) + wave * BNW;
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
  if (tid == 0) n_rows = 0;
  __syncthreads();
  if (tid < BM && row_id[tid] >= 0) atomicMax(&n_rows, tid + 1);
  __syncthreads();
  const int m_tiles = (n_rows + 15) >> 4;  // real rows are packed at the front of a block
  if (m_tiles == 0) return;

  // ---- A staging slots: chunk ch (16 B) of rows r0 + RPP*i ----
  const int ch = tid % CPRW, r0 = tid / CPRW;
  int64_t a_off[NA];  // element offset of the chunk at k = 0, -1 = no such row
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int rid = row_id[r0 + RPP * i];
    const int64_t arow = GROUPED ? (int64_t)(rid / top_k_div) : (int64_t)rid;
    a_off[i] = rid >= 0 ? arow * lda + ch * 8 : -1;
  }
  // ---- W rows of this wave: group g, lane = row c16, 16 bytes at k = q4*8 (+32 per k-step) ----
  bool w_ok[NG];
  const T* w_ptr[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    w_ok[g] = (n0 + g * 16 + c16) < N;
    w_ptr[g] = w + expert * N * K + (w_ok[g] ? (n0 + g * 16 + c16) : 0) * K + q4 * 8;
  }

  // split-K: blockIdx.z owns [z * chunks_per_split, (z + 1) * chunks_per_split) in units of 256 k
  const int64_t k_begin = (int64_t)blockIdx.z * chunks_per_split * 256;
  const int64_t k_end = min(K, k_begin + (int64_t)chunks_per_split * 256);

  FragS areg[NA];
  FragS wreg[2][NG][KSC];
  auto fetch_a = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      areg[i].u = make_uint4(0, 0, 0, 0);
      if (a_off[i] >= 0 && k0 + ch * 8 < k_end) areg[i].u = *reinterpret_cast<const uint4*>(a + a_off[i] + k0);
    }
  };
  auto fetch_w = [&](FragS (&r)[NG][KSC], int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        r[g][ks].u = make_uint4(0, 0, 0, 0);
        if (w_ok[g] && k0 + ks * 32 + q4 * 8 < k_end)
          r[g][ks].u = *reinterpret_cast<const uint4*>(w_ptr[g] + k0 + ks * 32);
      }
    }
  };
  auto stage_a = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<uint4*>(&a_lds[(r0 + RPP * i) * AS + ch * 8]) = areg[i].u;
  };

  f32x4 acc[NG][MT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint16_t* a_rd = &a_lds[c16 * AS + q4 * 8];

  auto compute = [&](FragS (&r)[NG][KSC]) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (t < m_tiles) {
          FragS b;
          b.u = *reinterpret_cast<const uint4*>(a_rd + t * 16 * AS + ks * 32);
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[g][t] = MfmaS<T>::mma(r[g][ks], b, acc[g][t]);
        }
      }
    }
  };

  fetch_a(k_begin);
  fetch_w(wreg[0], k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += 2 * KC) {
    // ---- even chunk ----
    __syncthreads();
    stage_a();
    __syncthreads();
    if (k0 + KC < k_end) {
      fetch_a(k0 + KC);
      fetch_w(wreg[1], k0 + KC);
    }
    compute(wreg[0]);
    // ---- odd chunk ----
    if (k0 + KC < k_end) {
      __syncthreads();
      stage_a();
      __syncthreads();
      if (k0 + 2 * KC < k_end) {
        fetch_a(k0 + 2 * KC);
        fetch_w(wreg[0], k0 + 2 * KC);
      }
      compute(wreg[1]);
    }
  }

  // ---- epilogue: lane holds C^T[n = n0 + g*16 + q4*4 + r][m = t*16 + c16] ----
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int64_t nb = n0 + g * 16 + q4 * 4;
    if (!GROUPED && gridDim.z > 1) {
      // split-K: fp32 partials [z][m_block * BM + row][N], summed in z order by splitk_reduce_kernel.
      // (An in-kernel "last workgroup reduces" needs agent-scope fences, which write back / invalidate
      // the whole XCD L2 per workgroup on gfx950: measured 10x slower than this second launch.)
      const int64_t rows_total = (int64_t)gridDim.y * BM;
      float* ws = partial_ws + ((int64_t)blockIdx.z * rows_total + m0) * N;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (t >= m_tiles || nb >= N) continue;
        float* dst = ws + (int64_t)(t * 16 + c16) * N + nb;
        if (nb + 4 <= N && (N % 4 == 0)) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[g][t][0], acc[g][t][1], acc[g][t][2], acc[g][t][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < N) dst[r] = acc[g][t][r];
        }
      }
      continue;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if (t >= m_tiles) continue;
      const int rid = row_id[t * 16 + c16];
      if (rid < 0 || nb >= N) continue;
      float v[4];
      const float scale = (GROUPED && mul_routed_weight) ? topk_weights[rid] : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[g][t][r] * scale;
      OutT* dst = c + (int64_t)rid * ldc + nb;
      if (nb + 4 <= N && (ldc % 4 == 0)) {
        if constexpr (sizeof(OutT) == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 p;
          p.x = (uint32_t)Elem<OutT>::from_f(v[0]).v | ((uint32_t)Elem<OutT>::from_f(v[1]).v << 16);
          p.y = (uint32_t)Elem<OutT>::from_f(v[2]).v | ((uint32_t)Elem<OutT>::from_f(v[3]).v << 16);
          *reinterpret_cast<uint2*>(dst) = p;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r < N) {
            if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float*>(dst + r) = v[r];
            else dst[r] = Elem<OutT>::from_f(v[r]);
          }
        }
      }
    }
  }
}

// usable when K is a multiple of 32 (whole MFMA k-steps) and rows are 16-byte aligned
bool skinny_gemm_ok(int64_t K, int64_t lda, const void* a, const void* w) {
  return K % 32 == 0 && lda % 8 == 0 && aligned16(a) && aligned16(w);
}

// c[m, n] = sum_z partial[z][m][n] (z ascending: deterministic), 4 consecutive n per thread
template <typename OutT>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(OutT* __restrict__ c, const float* __restrict__ partial, int ksplit, int64_t M, int64_t N,
                     int64_t plane_rows, int64_t ldc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = N / 4;
  if (i >= M * n4) return;
  const int64_t m = i / n4, n = (i - m * n4) * 4;
  const float* src = partial + m * N + n;
  f32x4 acc = *reinterpret_cast<const f32x4*>(src);
  for (int z = 1; z < ksplit; ++z) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)z * plane_rows * N);
  OutT* dst = c + m * ldc + n;
  if constexpr (sizeof(OutT) == 4) {
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    uint2 p;
    p.x = (uint32_t)Elem<OutT>::from_f(acc[0]).v | ((uint32_t)Elem<OutT>::from_f(acc[1]).v << 16);
    p.y = (uint32_t)Elem<OutT>::from_f(acc[2]).v | ((uint32_t)Elem<OutT>::from_f(acc[3]).v << 16);
    *reinterpret_cast<uint2*>(dst) = p;
  }
}

template <typename T, typename OutT, bool GROUPED, int BM = 64, int NG = (BM == 128 ? 4 : 1)>
int launch_skinny_gemm(OutT* c, const T* a, const T* w, const float* topk_weights, const int32_t* sorted_ids,
                       const int32_t* expert_ids, const int32_t* num_post_pad, int64_t num_valid, int64_t M,
                       int64_t N, int64_t K, int64_t lda, int64_t ldc, int64_t m_blocks, int top_k_div,
                       int mul_routed_weight, hipStream_t st, int ksplit, float* partial_ws) {
  const int chunks = (int)((K + 255) / 256);
  if (GROUPED || ksplit < 1 || !partial_ws || N % 4 != 0 || ldc % 4 != 0) ksplit = 1;
  const int cps = (chunks + ksplit - 1) / ksplit;
  ksplit = (chunks + cps - 1) / cps;  // no empty splits
  constexpr int64_t kRowsPerWg = 64 * NG;  // W rows per workgroup: 4 waves x NG groups of 16
  dim3 grid((unsigned)((N + kRowsPerWg - 1) / kRowsPerWg), (unsigned)m_blocks, (unsigned)ksplit);
  hipLaunchKernelGGL((skinny_gemm_kernel<T, OutT, GROUPED, BM, NG>), grid, dim3(256), 0, st, c, a, w, topk_weights,
                     sorted_ids, expert_ids, num_post_pad, num_valid, M, N, K, lda, ldc, top_k_div,
                     mul_routed_weight, cps, partial_ws);
  int rc = launch_status("skinny_gemm");
  if (rc || ksplit == 1) return rc;
  const int64_t items = M * (N / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<OutT>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, c,
                     (const float*)partial_ws, ksplit, M, N, m_blocks * BM, ldc);
  return launch_status("splitk_reduce");
}

// Split-K factor for a weight-streaming call on `num_cus` compute units: fill every workgroup slot
// (3 workgroups per CU at 152 VGPRs) in whole rounds, keep >= 2 K-chunks per split.
int skinny_pick_ksplit(int64_t rows, int64_t N, int64_t K, int64_t m_blocks, int num_cus, int rows_per_wg) {
  const int chunks = (int)((K + 255) / 256);
  const double tiles = (double)((N + rows_per_wg - 1) / rows_per_wg) * (double)m_blocks;
  const double slots = (double)(num_cus > 0 ? num_cus : 256) * 3.0;
  // time model: weight bytes at ~20 GB/s per resident workgroup slot in use (5 TB/s over a full chip),
  // scaled by how evenly the rounds fill the slots and the K-chunks fill the splits; a split adds
  // the reduction launch (~3 us) and its partial traffic
  const double bytes = (double)N * (double)K * 2.0;
  int best = 1;
  double best_t = 1e30;
  for (int s : {1, 2, 3, 4, 6, 8, 12, 16}) {
    if (s > 1 && chunks / s < 2) break;
    const int cps = (chunks + s - 1) / s;
    const int s_eff = (chunks + cps - 1) / cps;
    const double wgs = tiles * s_eff;
    const double rounds = wgs / slots;
    const double full_rounds = (double)(int64_t)(rounds + 0.999999);
    const double balance = (double)chunks / ((double)s_eff * cps);
    double t = bytes * full_rounds / (rounds * balance) / (20e9 * (rounds < 1.0 ? wgs : slots));
    if (s_eff > 1) t += 3e-6 + (double)s_eff * (double)rows * (double)N * 8.0 / 2e12;
    if (t < best_t * 0.97) {
      best_t = t;
      best = s_eff;
    }
  }
  return best;
}

#define SKINNY_INST(T, OutT, G, BM, NG)                                                                        \
  template int launch_skinny_gemm<T, OutT, G, BM, NG>(OutT*, const T*, const T*, const float*, const int32_t*, \
                                                      const int32_t*, const int32_t*, int64_t, int64_t, int64_t, \
                                                      int64_t, int64_t, int64_t, int64_t, int, int, hipStream_t, \
                                                      int, float*);
SKINNY_INST(bf16_t, bf16_t, true, 64, 1)
SKINNY_INST(f16_t, f16_t, true, 64, 1)
SKINNY_INST(bf16_t, bf16_t, true, 128, 4)
SKINNY_INST(f16_t, f16_t, true, 128, 4)
SKINNY_INST(bf16_t, float, false, 64, 1)
SKINNY_INST(f16_t, float, false, 64, 1)
SKINNY_INST(bf16_t, bf16_t, false, 64, 1)
SKINNY_INST(f16_t, f16_t, false, 64, 1)
SKINNY_INST(bf16_t, bf16_t, false, 64, 2)
SKINNY_INST(f16_t, f16_t, false, 64, 2)
SKINNY_INST(bf16_t, bf16_t, false, 64, 4)
SKINNY_INST(f16_t, f16_t, false, 64, 4)
#undef SKINNY_INST

}  // namespace semipd
