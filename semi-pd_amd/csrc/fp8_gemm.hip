// Block-scaled fp8 (OCP e4m3fn) linear algebra for gfx950:
//   * per_token_group_quant_fp8: activations -> fp8 + one fp32 scale per (row, group of K)
//     (layers/quantization/fp8_kernel.py:75-115, 165-250);
//   * w8a8_block_fp8_matmul: C[m, n] = sum_kb (sum_{k in block kb} Aq[m, k] * Wq[n, k]) * As[m, kb] * Ws[n / bn, kb]
//     (fp8_kernel.py:409-491, 694-800) and its grouped form inside fused_moe_kernel
//     (fused_moe_triton/fused_moe.py:174-243).
//
// The matmul is the weight-streaming kernel of skinny_gemm.hip with bytes instead of bf16: a wave owns
// NG groups of 16 weight rows and streams them from HBM straight into MFMA operand registers, one chunk
// of K ahead; the activation rows of the block (fp8) and their scales sit in LDS.  What changes:
//   * half the weight bytes per output, so the decode-sized blocks stage 512 bytes of K per row to keep
//     the same bytes in flight per lane (8 x 16 B, double buffered);
//   * one scale block of 128 k is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per tile (32 bytes per lane and
//     operand; hardware scales set to 2^0).  The non-scaled 16x16x32 fp8 MFMA runs at the bf16 rate on
//     gfx950, the K = 128 form at twice that (MI355X_MICROARCH.md).  A lane loads the 32 contiguous
//     bytes k = q4 * 32 .. + 32 of its weight row, the activation fragment is read from LDS at the same
//     offsets; operands A and B map lanes to k the same way, so the dot products are intact whatever the
//     instruction's order of k inside the block is;
//   * the block's partial tile is folded into the accumulator with As[m, kb] * Ws[n-block, kb] — one
//     multiply for the product of scales, four FMAs per tile (the fp32 scales of this format are not
//     powers of two, so the instruction's own E8M0 scaling cannot carry them);
//   * split-K for calls with few output tiles (fp32 partial planes + a reduce launch, as in skinny_gemm.hip).
#include "common.h"

#include <stdlib.h>

namespace semipd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr float kFp8Max = 448.0f;  // torch.finfo(torch.float8_e4m3fn).max; the reference's HIP branch uses
                                   // e4m3fnuz / 224 for MI300 (fp8_kernel.py:191-194), gfx950 is OCP

// ------------------------------------------------------------------------------------------------
// per-token-group quantisation: thread = 8 consecutive elements, a group = G / 8 consecutive lanes
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
per_token_group_quant_fp8_kernel(uint8_t* __restrict__ q, float* __restrict__ s, const T* __restrict__ x,
                                 int64_t num_vec, int lanes_per_group, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 8-element vector index
  const bool live = i < num_vec;
  float v[8];
  if (live) {
    if constexpr (sizeof(T) == 4) {
      const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
      const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(x + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = Elem<T>::to_f(a.e[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  // groups are aligned runs of lanes_per_group lanes (8, 16, 32 or 64) inside the wave
  for (int off = 1; off < lanes_per_group; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  amax = fmaxf(amax, eps);
  const float y_s = amax / kFp8Max;
  const float y_s_inv = 1.0f / y_s;
  if (!live) return;
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = fminf(fmaxf(v[j] * y_s_inv, -kFp8Max), kFp8Max);
  uint2 p;
  p.x = F8Cvt<f8e4m3_t>::pack2<false>(w[0], w[1], 0u);
  p.x = F8Cvt<f8e4m3_t>::pack2<true>(w[2], w[3], p.x);
  p.y = F8Cvt<f8e4m3_t>::pack2<false>(w[4], w[5], 0u);
  p.y = F8Cvt<f8e4m3_t>::pack2<true>(w[6], w[7], p.y);
  *reinterpret_cast<uint2*>(q + i * 8) = p;
  if ((threadIdx.x & (lanes_per_group - 1)) == 0) s[i / lanes_per_group] = y_s;
}

// SiLU(gate) * up and the quantisation of the product in one pass: x [rows, 2 d] -> q [rows, d], s [rows, d / G].
// The product is rounded to T first, as SiluAndMul's output is (layers/activation.py:41-44), so the bytes equal
// those of silu_and_mul followed by per_token_group_quant_fp8 (fused_moe.py:1104-1125 runs them back to back);
// the T-typed intermediate never travels through HBM.
template <typename T>
__global__ void __launch_bounds__(256)
silu_and_mul_quant_fp8_kernel(uint8_t* __restrict__ q, float* __restrict__ s, const T* __restrict__ x, int64_t num_vec,
                              int vec_per_row, int lanes_per_group, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 8-element output vector index
  const bool live = i < num_vec;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (live) {
    const int64_t row = i / vec_per_row;
    const int c = (int)(i - row * vec_per_row);
    const T* xr = x + row * (int64_t)vec_per_row * 16;
    const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(xr + (int64_t)c * 8);
    const Vec16<T> u = *reinterpret_cast<const Vec16<T>*>(xr + (int64_t)(vec_per_row + c) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gf = Elem<T>::to_f(g.e[j]);
      v[j] = Elem<T>::to_f(Elem<T>::from_f(gf / (1.0f + __expf(-gf)) * Elem<T>::to_f(u.e[j])));
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
  for (int off = 1; off < lanes_per_group; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  amax = fmaxf(amax, eps);
  const float y_s = amax / kFp8Max;
  const float y_s_inv = 1.0f / y_s;
  if (!live) return;
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = fminf(fmaxf(v[j] * y_s_inv, -kFp8Max), kFp8Max);
  uint2 p;
  p.x = F8Cvt<f8e4m3_t>::pack2<false>(w[0], w[1], 0u);
  p.x = F8Cvt<f8e4m3_t>::pack2<true>(w[2], w[3], p.x);
  p.y = F8Cvt<f8e4m3_t>::pack2<false>(w[4], w[5], 0u);
  p.y = F8Cvt<f8e4m3_t>::pack2<true>(w[6], w[7], p.y);
  *reinterpret_cast<uint2*>(q + i * 8) = p;
  if ((threadIdx.x & (lanes_per_group - 1)) == 0) s[i / lanes_per_group] = y_s;
}

// ------------------------------------------------------------------------------------------------
// block-scaled fp8 NT GEMM (dense and grouped)
// ------------------------------------------------------------------------------------------------
template <typename OutT, bool GROUPED, int BM, int NG, int KC>
__global__ void __launch_bounds__(256, 2)
fp8_block_gemm_kernel(OutT* __restrict__ c, const uint8_t* __restrict__ a, const float* __restrict__ a_s,
                      const uint8_t* __restrict__ w, const float* __restrict__ w_s,
                      const float* __restrict__ topk_weights, const int32_t* __restrict__ sorted_ids,
                      const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad,
                      int64_t num_valid, int64_t M, int64_t N, int64_t K, int64_t ldc, int block_n, int top_k_div,
                      int mul_routed_weight, int chunks_per_split, float* __restrict__ partial_ws, int n_tiles, int m_blocks,
                      int group_n) {
  static_assert(NG == 1 || NG == 2 || NG == 4, "NG = groups of 16 W rows per wave");
  // KC = bytes of K staged per barrier pair: 512 for the 64-row blocks (8 x 16 B of weights in flight per
  // lane, double buffered), 128 for the 128-row blocks (their 64 accumulator registers leave room for less)
  static_assert(KC == 128 || KC == 256 || KC == 512, "whole scale blocks per chunk");
  constexpr int SB = KC / 128;     // scale blocks per chunk
  constexpr int SS = KC / 64;      // 16-byte loads per lane and W row per chunk (two per scale block)
  static_assert(SS == 2 * SB, "two 16-byte halves per scale block");
  constexpr int BNW = 16 * NG;     // W rows per wave
  constexpr int MT = BM / 16;      // m-tiles
  constexpr int AS = KC + 16;      // LDS row stride in bytes
  constexpr int CPRW = KC / 16;    // 16-byte chunks per staged row
  constexpr int RPP = 256 / CPRW;  // rows staged per pass of the 256 threads
  constexpr int NA = BM / RPP;     // 16-byte activation chunks per thread per K-chunk
  static_assert(BM * SB <= 256, "one scale per thread");
  __shared__ __attribute__((aligned(16))) uint8_t a_lds[BM * AS];
  __shared__ float as_lds[BM * SB];
  __shared__ int row_id[BM];
  __shared__ int n_rows;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  int tile_n = blockIdx.x, tile_m = blockIdx.y;
  if (!GROUPED) {
    // Dense calls come as a 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch order; a wrong guess costs
    // speed, not correctness), and each XCD has its own 4 MB L2: give every XCD a contiguous range of tiles
    // (bijective remap of cdna_hip_programming.md) and walk it as "group_n weight tiles x all row blocks", so the
    // group's weight tiles stay in that L2 while the activation tiles stream past.
    const int nwg = n_tiles * m_blocks;
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int per_group = group_n * m_blocks;
    const int grp = logical / per_group, within = logical - grp * per_group;
    const int gn = min(group_n, n_tiles - grp * group_n);  // the last group may be narrower
    tile_n = grp * group_n + within % gn;
    tile_m = within / gn;
  }
  const int64_t m0 = (int64_t)tile_m * BM;
  const int64_t n0 = (int64_t)tile_n * (4 * BNW) + wave * BNW;
  const int64_t KB = (K + 127) / 128;  // scale blocks along K
  int64_t expert = 0;
  if (GROUPED) {
    if (m0 >= *num_post_pad) return;
    expert = expert_ids[tile_m];
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
  const int m_tiles = (n_rows + 15) >> 4;
  if (m_tiles == 0) return;

  const int ch = tid % CPRW, r0 = tid / CPRW;
  int64_t a_off[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int rid = row_id[r0 + RPP * i];
    const int64_t arow = GROUPED ? (int64_t)(rid / top_k_div) : (int64_t)rid;
    a_off[i] = rid >= 0 ? arow * K + ch * 16 : -1;
  }
  // activation scale slot of this thread: row tid / SB, scale block tid % SB of the chunk
  int64_t s_off = -1;
  const int s_j = tid % SB;
  if (tid < BM * SB) {
    const int rid = row_id[tid / SB];
    const int64_t arow = GROUPED ? (int64_t)(rid / top_k_div) : (int64_t)rid;
    if (rid >= 0) s_off = arow * KB + s_j;
  }
  bool w_ok[NG];
  const uint8_t* w_ptr[NG];
  const float* ws_ptr[NG];
  const int64_t NB = (N + block_n - 1) / block_n;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    w_ok[g] = (n0 + g * 16 + c16) < N;
    w_ptr[g] = w + expert * N * K + (w_ok[g] ? (n0 + g * 16 + c16) : 0) * K + q4 * 32;
    const int64_t nb = (n0 + g * 16 < N ? n0 + g * 16 : 0) / block_n;  // one scale row per group of 16 (16 | block_n)
    ws_ptr[g] = w_s + (expert * NB + nb) * KB;
  }

  // split-K: blockIdx.z owns chunks [z * chunks_per_split, (z + 1) * chunks_per_split) of KC bytes
  const int64_t k_begin = (int64_t)blockIdx.z * chunks_per_split * KC;
  const int64_t k_end = min(K, k_begin + (int64_t)chunks_per_split * KC);

  uint4 areg[NA];
  float sreg = 0.f;
  i32x8 wreg[2][NG][SB];  // one MFMA operand (32 bytes) per scale block
  float wsreg[2][NG][SB];  // and its weight scale, fetched with it (a load inside the compute phase would
                           // have to wait for the whole prefetch: vmcnt counts in order)
  auto fetch_a = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      areg[i] = make_uint4(0, 0, 0, 0);
      if (a_off[i] >= 0 && k0 + ch * 16 < k_end) areg[i] = *reinterpret_cast<const uint4*>(a + a_off[i] + k0);
    }
    sreg = 0.f;
    if (s_off >= 0 && k0 / 128 + s_j < KB) sreg = a_s[s_off + k0 / 128];
  };
  auto fetch_w = [&](i32x8 (&r)[NG][SB], float (&rs)[NG][SB], int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        rs[g][sb] = (k0 + sb * 128 < k_end) ? ws_ptr[g][k0 / 128 + sb] : 0.f;
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int sb = 0; sb < SB; ++sb) {
        const int64_t kk = k0 + sb * 128;  // + q4 * 32 in w_ptr; K % 16 == 0, so the halves are guarded separately
        i32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (w_ok[g] && kk + q4 * 32 < k_end) lo = *reinterpret_cast<const i32x4*>(w_ptr[g] + kk);
        if (w_ok[g] && kk + q4 * 32 + 16 < k_end) hi = *reinterpret_cast<const i32x4*>(w_ptr[g] + kk + 16);
        r[g][sb].lo = lo;
        r[g][sb].hi = hi;
      }
    }
  };
  auto stage_a = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(&a_lds[(r0 + RPP * i) * AS + ch * 16]) = areg[i];
    if (tid < BM * SB) as_lds[tid] = sreg;
  };

  f32x4 acc[NG][MT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint8_t* a_rd = &a_lds[c16 * AS + q4 * 32];

  // One scale block: MFMA of tile t + 1 is in flight while tile t is folded into the accumulators.  TILES is
  // a compile-time count (all of them, or half for blocks that are at most half full): no per-tile branches,
  // rows of absent tokens are zero in LDS.
#define SEMIPD_FP8_BLOCK_STEP(NAME, TILES)                                                                             \
  auto NAME = [&](i32x8 (&r)[NG][SB], float (&rs)[NG][SB], int sb) __attribute__((always_inline)) {                    \
    f32x4 blk[2][NG];                                                                                                  \
    _Pragma("unroll") for (int t = -1; t < (TILES); ++t) {                                                             \
      if (t + 1 < (TILES)) {                                                                                           \
        i32x8 bv;                                                                                                      \
        bv.lo = *reinterpret_cast<const i32x4*>(a_rd + (t + 1) * 16 * AS + sb * 128);                                  \
        bv.hi = *reinterpret_cast<const i32x4*>(a_rd + (t + 1) * 16 * AS + sb * 128 + 16);                             \
        _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                               \
          /* cbsz = blgp = 0: both operands e4m3; hardware scales 0x7f = 2^0 */                                        \
          blk[(t + 1) & 1][g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                                      \
              r[g][sb], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);                            \
        }                                                                                                              \
      }                                                                                                                \
      if (t >= 0) {                                                                                                    \
        const float asc = as_lds[(t * 16 + c16) * SB + sb];                                                            \
        _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                               \
          const float sc = asc * rs[g][sb];                                                                            \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[g][t][e] += blk[t & 1][g][e] * sc;                         \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }
  SEMIPD_FP8_BLOCK_STEP(block_step_full, MT);
  SEMIPD_FP8_BLOCK_STEP(block_step_half, MT / 2);
#undef SEMIPD_FP8_BLOCK_STEP
  auto compute = [&](i32x8 (&r)[NG][SB], float (&rs)[NG][SB], int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int sb = 0; sb < SB; ++sb) {
      if (k0 + sb * 128 >= k_end) break;
      if (m_tiles * 2 <= MT) block_step_half(r, rs, sb);
      else block_step_full(r, rs, sb);
    }
  };

  fetch_a(k_begin);
  fetch_w(wreg[0], wsreg[0], k_begin);
  for (int64_t k0 = k_begin; k0 < k_end; k0 += 2 * KC) {
    __syncthreads();
    stage_a();
    __syncthreads();
    if (k0 + KC < k_end) {
      fetch_a(k0 + KC);
      fetch_w(wreg[1], wsreg[1], k0 + KC);
    }
    compute(wreg[0], wsreg[0], k0);
    if (k0 + KC < k_end) {
      __syncthreads();
      stage_a();
      __syncthreads();
      if (k0 + 2 * KC < k_end) {
        fetch_a(k0 + 2 * KC);
        fetch_w(wreg[0], wsreg[0], k0 + 2 * KC);
      }
      compute(wreg[1], wsreg[1], k0 + KC);
    }
  }

  // ---- epilogue: lane holds C^T[n = n0 + g*16 + q4*4 + r][m = t*16 + c16] ----
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int64_t nb = n0 + g * 16 + q4 * 4;
    if (!GROUPED && gridDim.z > 1) {
      // fp32 partials [z][m_block * BM + row][N], summed in z order by fp8_splitk_reduce_kernel
      const int64_t rows_total = (int64_t)m_blocks * BM;
      float* ws = partial_ws + ((int64_t)blockIdx.z * rows_total + m0) * N;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (t >= m_tiles || nb >= N) continue;
        float* dst = ws + (int64_t)(t * 16 + c16) * N + nb;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[g][t][0], acc[g][t][1], acc[g][t][2], acc[g][t][3]);
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

template <typename OutT>
__global__ void __launch_bounds__(256)
fp8_splitk_reduce_kernel(OutT* __restrict__ c, const float* __restrict__ partial, int ksplit, int64_t M, int64_t N,
                         int64_t plane_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = N / 4;
  if (i >= M * n4) return;
  const int64_t m = i / n4, n = (i - m * n4) * 4;
  const float* src = partial + m * N + n;
  f32x4 acc = *reinterpret_cast<const f32x4*>(src);
  for (int z = 1; z < ksplit; ++z) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)z * plane_rows * N);
  OutT* dst = c + m * N + n;
  if constexpr (sizeof(OutT) == 4) {
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    uint2 p;
    p.x = (uint32_t)Elem<OutT>::from_f(acc[0]).v | ((uint32_t)Elem<OutT>::from_f(acc[1]).v << 16);
    p.y = (uint32_t)Elem<OutT>::from_f(acc[2]).v | ((uint32_t)Elem<OutT>::from_f(acc[3]).v << 16);
    *reinterpret_cast<uint2*>(dst) = p;
  }
}

static long env_int(const char* name, long dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atol(v) : dflt;
}

// Split-K factor: enough workgroups for two per CU (the kernel's occupancy), at least two chunks per split,
// and only when the fp32 partial planes fit the workspace.
static int fp8_pick_ksplit(int64_t tiles, int chunks, int64_t rows_total, int64_t N, size_t ws_bytes) {
  int s = 1;
  while (s < 16 && tiles * s < 512 && chunks / (s * 2) >= 2) s *= 2;
  while (s > 1 && (size_t)s * rows_total * N * 4 > ws_bytes) s /= 2;
  return s;
}

template <typename OutT, bool GROUPED, int BM, int NG, int KC>
static int launch_fp8_gemm(void* c, const void* a, const float* a_s, const void* w, const float* w_s,
                           const float* topk_weights, const int32_t* sorted_ids, const int32_t* expert_ids,
                           const int32_t* num_post_pad, int64_t num_valid, int64_t M, int64_t N, int64_t K, int64_t ldc,
                           int64_t m_blocks, int block_n, int top_k_div, int mul_routed_weight, hipStream_t st,
                           float* ws, size_t ws_bytes) {
  constexpr int64_t kRowsPerWg = 64 * NG;
  const int64_t n_tiles = (N + kRowsPerWg - 1) / kRowsPerWg;
  const int chunks = (int)((K + KC - 1) / KC);
  int ksplit = 1;
  if (!GROUPED && ws && N % 4 == 0) ksplit = fp8_pick_ksplit(n_tiles * m_blocks, chunks, m_blocks * BM, N, ws_bytes);
  if (const long forced = env_int("SEMIPD_FP8_KSPLIT", 0); forced > 0 && !GROUPED && ws && N % 4 == 0) ksplit = (int)forced;
  const int cps = (chunks + ksplit - 1) / ksplit;
  ksplit = (chunks + cps - 1) / cps;  // no empty splits
  dim3 grid = GROUPED ? dim3((unsigned)n_tiles, (unsigned)m_blocks, 1)
                      : dim3((unsigned)(n_tiles * m_blocks), 1, (unsigned)ksplit);
  // weight tiles per group: measured on DeepSeek-V3 shapes at M = 4096, 8 is best (1.25 vs 1.09 PFLOP/s with 1;
  // profiles/r01_kbench_fp8_v1.txt) — a mild effect: the 256 MB MALL already holds the whole weight matrix
  int group_n = (int)env_int("SEMIPD_FP8_GROUP_N", 8);
  if (group_n < 1) group_n = 1;
  hipLaunchKernelGGL((fp8_block_gemm_kernel<OutT, GROUPED, BM, NG, KC>), grid, dim3(256), 0, st, (OutT*)c, (const uint8_t*)a,
                     a_s, (const uint8_t*)w, w_s, topk_weights, sorted_ids, expert_ids, num_post_pad, num_valid, M, N, K,
                     ldc, block_n, top_k_div, mul_routed_weight, cps, ws, (int)n_tiles, (int)m_blocks, group_n);
  int rc = launch_status("fp8_block_gemm");
  if (rc || ksplit == 1) return rc;
  const int64_t items = M * (N / 4);
  hipLaunchKernelGGL((fp8_splitk_reduce_kernel<OutT>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (OutT*)c,
                     (const float*)ws, ksplit, M, N, m_blocks * BM);
  return launch_status("fp8_splitk_reduce");
}

template <bool GROUPED>
static int dispatch_fp8_gemm(void* c, const void* a, const float* a_s, const void* w, const float* w_s,
                             const float* topk_weights, const int32_t* sorted_ids, const int32_t* expert_ids,
                             const int32_t* num_post_pad, int64_t num_valid, int64_t M, int64_t N, int64_t K, int64_t ldc,
                             int64_t m_blocks, int block_m, int block_n, int top_k_div, int mul_routed_weight,
                             int out_dtype, hipStream_t st, float* ws, size_t ws_bytes) {
#define GO(OutT)                                                                                                       \
  do {                                                                                                                 \
    if (block_m == 128)                                                                                                \
      return launch_fp8_gemm<OutT, GROUPED, 128, 2, 128>(c, a, a_s, w, w_s, topk_weights, sorted_ids, expert_ids,          \
                                                    num_post_pad, num_valid, M, N, K, ldc, m_blocks, block_n, top_k_div, \
                                                    mul_routed_weight, st, ws, ws_bytes);                              \
    return launch_fp8_gemm<OutT, GROUPED, 64, 1, 512>(c, a, a_s, w, w_s, topk_weights, sorted_ids, expert_ids, num_post_pad, \
                                                 num_valid, M, N, K, ldc, m_blocks, block_n, top_k_div,                \
                                                 mul_routed_weight, st, ws, ws_bytes);                                 \
  } while (0)
  switch (out_dtype) {
    case SEMIPD_BF16: GO(bf16_t);
    case SEMIPD_F16: GO(f16_t);
    case SEMIPD_F32: GO(float);
  }
#undef GO
  set_error("block fp8 matmul: output dtype code %d is not f32 / bf16 / f16", out_dtype);
  return SEMIPD_EDTYPE;
}

static int check_fp8_gemm_args(const char* what, int64_t N, int64_t K, int block_n, int block_k, const void* a,
                               const void* w) {
  SEMIPD_CHECK_ARG(block_k == 128, SEMIPD_ESHAPE, "%s: block_k must be 128 (got %d)", what, block_k);
  SEMIPD_CHECK_ARG(block_n > 0 && block_n % 16 == 0, SEMIPD_ESHAPE, "%s: block_n must be a multiple of 16 (got %d)", what,
                   block_n);
  SEMIPD_CHECK_ARG(N > 0 && K > 0 && K % 16 == 0, SEMIPD_ESHAPE, "%s: K must be a positive multiple of 16 (N=%lld K=%lld)",
                   what, (long long)N, (long long)K);
  SEMIPD_CHECK_ARG(aligned16(a) && aligned16(w), SEMIPD_EALIGN, "%s: operands must be 16-byte aligned", what);
  return 0;
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_per_token_group_quant_fp8(void* q, float* s, const void* x, int64_t num_rows, int64_t hidden,
                                     int group_size, float eps, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(q && s && x, SEMIPD_EINVAL, "per_token_group_quant_fp8: null pointer");
  SEMIPD_CHECK_ARG(group_size == 64 || group_size == 128 || group_size == 256 || group_size == 512, SEMIPD_ESHAPE,
                   "per_token_group_quant_fp8: group size %d is not one of 64, 128, 256, 512", group_size);
  // fp8_kernel.py:183-186: "the last dimension of `x` cannot be divisible by `group_size`"
  SEMIPD_CHECK_ARG(hidden > 0 && hidden % group_size == 0, SEMIPD_ESHAPE,
                   "per_token_group_quant_fp8: hidden size %lld is not a multiple of the group size %d", (long long)hidden,
                   group_size);
  SEMIPD_CHECK_ARG(aligned16(x) && (reinterpret_cast<uintptr_t>(q) & 7u) == 0, SEMIPD_EALIGN,
                   "per_token_group_quant_fp8: x must be 16-byte and q 8-byte aligned");
  if (num_rows == 0) return 0;
  const int64_t num_vec = num_rows * hidden / 8;
  dim3 grid((unsigned)((num_vec + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpg = group_size / 8;
  switch (dtype) {
    case SEMIPD_F32:
      hipLaunchKernelGGL((per_token_group_quant_fp8_kernel<float>), grid, dim3(256), 0, st, (uint8_t*)q, s, (const float*)x,
                         num_vec, lpg, eps);
      break;
    case SEMIPD_BF16:
      hipLaunchKernelGGL((per_token_group_quant_fp8_kernel<bf16_t>), grid, dim3(256), 0, st, (uint8_t*)q, s,
                         (const bf16_t*)x, num_vec, lpg, eps);
      break;
    case SEMIPD_F16:
      hipLaunchKernelGGL((per_token_group_quant_fp8_kernel<f16_t>), grid, dim3(256), 0, st, (uint8_t*)q, s, (const f16_t*)x,
                         num_vec, lpg, eps);
      break;
    default:
      set_error("per_token_group_quant_fp8: dtype code %d is not f32 / bf16 / f16", dtype);
      return SEMIPD_EDTYPE;
  }
  return launch_status("per_token_group_quant_fp8");
}

int semipd_silu_and_mul_quant_fp8(void* q, float* s, const void* x, int64_t num_rows, int64_t d, int group_size,
                                  float eps, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(q && s && x, SEMIPD_EINVAL, "silu_and_mul_quant_fp8: null pointer");
  SEMIPD_CHECK_ARG(group_size == 64 || group_size == 128 || group_size == 256 || group_size == 512, SEMIPD_ESHAPE,
                   "silu_and_mul_quant_fp8: group size %d is not one of 64, 128, 256, 512", group_size);
  SEMIPD_CHECK_ARG(d > 0 && d % group_size == 0, SEMIPD_ESHAPE,
                   "silu_and_mul_quant_fp8: width %lld is not a multiple of the group size %d", (long long)d, group_size);
  SEMIPD_CHECK_ARG(aligned16(x) && (reinterpret_cast<uintptr_t>(q) & 7u) == 0, SEMIPD_EALIGN,
                   "silu_and_mul_quant_fp8: x must be 16-byte and q 8-byte aligned");
  if (num_rows == 0) return 0;
  const int64_t num_vec = num_rows * d / 8;
  dim3 grid((unsigned)((num_vec + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpg = group_size / 8, vpr = (int)(d / 8);
  switch (dtype) {
    case SEMIPD_BF16:
      hipLaunchKernelGGL((silu_and_mul_quant_fp8_kernel<bf16_t>), grid, dim3(256), 0, st, (uint8_t*)q, s, (const bf16_t*)x,
                         num_vec, vpr, lpg, eps);
      break;
    case SEMIPD_F16:
      hipLaunchKernelGGL((silu_and_mul_quant_fp8_kernel<f16_t>), grid, dim3(256), 0, st, (uint8_t*)q, s, (const f16_t*)x,
                         num_vec, vpr, lpg, eps);
      break;
    default:
      set_error("silu_and_mul_quant_fp8: dtype code %d is not bf16 / f16", dtype);
      return SEMIPD_EDTYPE;
  }
  return launch_status("silu_and_mul_quant_fp8");
}

int semipd_w8a8_block_fp8_matmul(void* c, const void* a_q, const float* a_s, const void* w_q, const float* w_s,
                                 int64_t m, int64_t n, int64_t k, int block_n, int block_k, int out_dtype,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  SEMIPD_CHECK_ARG(c && a_q && a_s && w_q && w_s, SEMIPD_EINVAL, "w8a8_block_fp8_matmul: null pointer");
  if (int rc = check_fp8_gemm_args("w8a8_block_fp8_matmul", n, k, block_n, block_k, a_q, w_q)) return rc;
  if (m == 0) return 0;
  // 64-row blocks also for up to 512 rows when the weight matrix is small (<= 20 M elements: it stays in L2 / MALL
  // and more, smaller workgroups fill the chip better: 35 vs 44 us at M = 512, 7168 x 2048); big matrices keep the
  // 128-row blocks, which read the weights half as often (profiles/r01_kbench_fp8_v1.txt)
  const long small_m_upto = env_int("SEMIPD_FP8_BM64_UPTO", (n * k <= 20000000) ? 512 : 64);
  const int block_m = m <= small_m_upto ? 64 : 128;
  const int64_t m_blocks = (m + block_m - 1) / block_m;
  return dispatch_fp8_gemm<false>(c, a_q, a_s, w_q, w_s, nullptr, nullptr, nullptr, nullptr, 0, m, n, k, n, m_blocks, block_m,
                                  block_n, 1, 0, out_dtype, static_cast<hipStream_t>(stream), static_cast<float*>(workspace),
                                  workspace ? workspace_bytes : 0);
}

int semipd_moe_grouped_gemm_fp8(void* c, const void* a_q, const float* a_s, const void* w_q, const float* w_s,
                                const float* topk_weights, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n, int64_t k,
                                int64_t max_sorted, int top_k_div, int mul_routed_weight, int block_m, int block_n,
                                int block_k, int out_dtype, void* stream) {
  SEMIPD_CHECK_ARG(c && a_q && a_s && w_q && w_s && sorted_token_ids && expert_ids && num_tokens_post_pad, SEMIPD_EINVAL,
                   "moe_grouped_gemm_fp8: null pointer");
  SEMIPD_CHECK_ARG(!mul_routed_weight || topk_weights, SEMIPD_EINVAL, "moe_grouped_gemm_fp8: routed weights missing");
  SEMIPD_CHECK_ARG(block_m == 64 || block_m == 128, SEMIPD_ESHAPE, "moe_grouped_gemm_fp8: block_m must be 64 or 128");
  SEMIPD_CHECK_ARG(top_k_div >= 1 && max_sorted >= 0, SEMIPD_ESHAPE, "moe_grouped_gemm_fp8: top_k_div %d, max_sorted %lld",
                   top_k_div, (long long)max_sorted);
  if (int rc = check_fp8_gemm_args("moe_grouped_gemm_fp8", n, k, block_n, block_k, a_q, w_q)) return rc;
  if (num_valid == 0 || max_sorted == 0) return 0;
  return dispatch_fp8_gemm<true>(c, a_q, a_s, w_q, w_s, topk_weights, sorted_token_ids, expert_ids, num_tokens_post_pad,
                                 num_valid, 0, n, k, n, (max_sorted + block_m - 1) / block_m, block_m, block_n, top_k_div,
                                 mul_routed_weight,
                                 out_dtype, static_cast<hipStream_t>(stream), nullptr, 0);
}

}  // extern "C"
