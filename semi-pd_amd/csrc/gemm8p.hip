// out[m, n] = sum_k x[m, k] * W[n, k] for batches that are too tall for the weight-streaming kernel (65 rows and up):
// a 256 x 256 output tile per 8-wave workgroup, K in steps of 64, both operands through LDS-DMA, the two wave groups
// of a workgroup half a phase apart so that one group's MFMAs cover the other's LDS reads (ping-pong).
//
// Tile anatomy (all sizes in rows x k):
//   * the x tile (256 x 64) and the W tile (256 x 64) of a K step are each staged as TWO half-tiles of 128 rows.  A
//     half-tile is not a contiguous half: half 0 of x holds the rows every wave needs FIRST (rows 0-63 of both 128-row
//     wave bands), half 1 the rows needed later (64-127 of both bands); half 0 of W holds rows 0-31 of each of the four
//     64-row wave columns, half 1 rows 32-63.  So the four phases of a K step read  x0 + W0 | W1 | x1 | (nothing),
//     a half-tile's LDS slot is free again one phase later, and its successor (two K steps ahead) can be in flight for
//     five to six phases -- about 1.1 us of matrix work, which is what a load from HBM needs.
//   * LDS: 2 parities x 4 half-tiles x 16 KB = 128 KB.  Images are lane-linear (what the DMA writes): a wave
//     instruction fills 8 rows x 128 B, and the bank-conflict swizzle sits on the SOURCE address: 16-byte chunk c of row
//     r lands at position c ^ ((r >> 1) & 7); a ds_read_b128 of an MFMA fragment (16 rows, one chunk) then touches 16
//     distinct 16-byte slots.
//   * wave (wr, wc) of the 2 x 4 grid owns out rows 128 wr .. +127, columns 64 wc .. +63: 8 x 4 accumulator tiles of
//     v_mfma_f32_16x16x32 in the transposed form (MFMA A = W fragment, B = x fragment), so a lane ends with 4
//     consecutive columns of one row (8-byte stores).  Per K step and wave 64 MFMAs in four phases of 16 (one
//     64 x 32 quadrant each); fragment registers: x (32) + W lo (16) + W hi (16).
//   * phase = { ds_read this phase's fragments; issue one half-tile's share of DMA (2 instructions per wave); counted
//     s_waitcnt vmcnt; s_barrier; lgkmcnt(0); 16 MFMAs at raised priority; s_barrier }.  Waves with wr = 1 pass one extra
//     barrier up front (wr = 0 one at the end), which puts the two groups half a phase apart: while one group multiplies
//     the other reads LDS and issues DMA.  The two groups share the SIMDs pairwise (wave w and w + 4).
//   * ordering of LDS-DMA data: a wave waits (counted vmcnt) for ITS share of a half-tile before a barrier that every
//     reader passes before its ds_read; a slot is re-filled at least two phases after its last read.
// Epilogues: plain store, SiLU(gate) * up for a merged [gate; up] weight (W half 0 = gate rows, half 1 = the up rows of
// the same columns), fp32 K-slice planes for split-K.
// replaces UnquantizedLinearMethod.apply -> F.linear (+ SiluAndMul) (layers/linear.py:165-172, models/llama.py:88-92,
// layers/activation.py:41-53) for 65 .. 256-row decode batches, and _get_logits (layers/logits_processor.py:394-445).
#include "common.h"

namespace semipd {

typedef float g8_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g8_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g8_f16x8 __attribute__((ext_vector_type(8)));

union G8Frag {
  uint4 u;
  g8_bf16x8 b;
  g8_f16x8 f;
};
template <typename T> struct G8Mfma;
template <> struct G8Mfma<bf16_t> {
  __device__ static inline g8_f32x4 mma(const G8Frag& a, const G8Frag& b, g8_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct G8Mfma<f16_t> {
  __device__ static inline g8_f32x4 mma(const G8Frag& a, const G8Frag& b, g8_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
  }
};

enum { G8_PLAIN = 0, G8_SILU_MUL = 1 };

#define G8_GLDS(gp, lp, aux) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), \
                                   (__attribute__((address_space(3))) void*)(lp), 16, 0, aux)

__device__ inline uint4 g8_lds_read16(uint32_t addr, int imm) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm));
  return v;
}
template <int N> __device__ inline void g8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void g8_wait_lgkm0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ inline void g8_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Geometry of a tile: XH = rows of an x half-tile (128: a 256 x 256 tile for batches above 128 rows; 64: a 128 x 512
// tile for 65 .. 128 rows -- the same 32 accumulator tiles and 16 MFMAs per phase per wave, twice the weight rows per
// MFMA, so the short batches are not bound by matrix work spent on padding rows).
template <int XH> struct G8Geo {
  static constexpr int WH = 16384 / XH;                 // rows of a W half-tile: 128 | 256
  static constexpr int kXHalf = XH * 128, kWHalf = WH * 128;
  static constexpr int kParity = 2 * (kXHalf + kWHalf); // x0, x1, W0, W1 of one K-step parity: 64 KB | 80 KB
  static constexpr int kLds = 2 * kParity;              // 128 KB | 160 KB
  static constexpr int LX = XH / 64, LW = WH / 64;      // DMA instructions per wave per half-tile
  static constexpr int MT2 = XH / 32, NT2 = WH / 64;    // 16-row fragments per wave and half: x | W
  static constexpr int BM = 2 * XH, BN = 2 * WH;
  static constexpr int kInFlight = 2 * LX + 2 * LW;     // loads younger than the half-tile a wait is for
};

// NT: the weight tile is used by this workgroup only (one row of tiles): stream it non-temporally past L2 / MALL
// Grouped form (GR; fused-MoE expert GEMMs, fused_moe.py:54-273): the x tile is 2 XH consecutive entries of
// sorted_token_ids (moe_align_block_size with block size 2 XH: one expert per tile, expert_ids[tile]); entry id reads
// activation row id / top_k_div and writes output row id (entries >= num_valid are padding), optionally scaled by
// topk_weights[id]; W = w[expert].
struct G8Group {
  const int32_t* sorted_ids;
  const int32_t* expert_ids;
  const int32_t* num_post_pad;
  const float* topk_weights;
  int num_valid, top_k_div, mul_routed_weight;
};

template <typename T, int EPI, bool NT, int XH, bool GR = false, int ORDER = 0>
__global__ void __launch_bounds__(512)
gemm8p_kernel(T* __restrict__ out, float* __restrict__ planes, const T* __restrict__ x, const T* __restrict__ w, int M, int N,
              int K, int64_t ldx, int64_t ldo, int kt_per_slice, G8Group grp = G8Group()) {
  using G = G8Geo<XH>;
  constexpr int WH = G::WH, LX = G::LX, LW = G::LW, MT2 = G::MT2, NT2 = G::NT2;
  extern __shared__ __attribute__((aligned(16))) char g8_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int n_cols = EPI == G8_SILU_MUL ? N / 2 : N;       // output columns
  constexpr int cols_per_tile = EPI == G8_SILU_MUL ? WH : G::BN;
  const int tiles_n = (n_cols + cols_per_tile - 1) / cols_per_tile;
  int tile_m, tile_n;
  if (GR || ORDER == 0) {
    tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;   // neighbours share the x tile
  } else {
    // several row tiles (prefill-sized calls): workgroup id runs on XCD id % 8; give every XCD a contiguous range of
    // logical tiles and walk the row tiles fastest, so that the row tiles of one W tile sit on ONE XCD next to each
    // other in time: the W tile is fetched once into that L2 instead of once per row tile from the fabric
    const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    const int tiles_m = nwg / tiles_n;
    tile_n = logical / tiles_m, tile_m = logical - tile_n * tiles_m;
  }
  const int m0 = tile_m * G::BM, n0 = tile_n * cols_per_tile;
  const int ks = blockIdx.y;
  const int nkt_total = K >> 6;
  const int kt0 = ks * kt_per_slice;
  const int nkt = min(kt_per_slice, nkt_total - kt0);     // >= 1 by construction of the grid
  if (GR) {
    if (m0 >= grp.num_post_pad[0]) return;                // whole workgroup: tiles past the padded token count
    w += (int64_t)grp.expert_ids[tile_m] * N * K;
  }

  // ---- DMA sources.  A half-tile of R rows is R / 8 wave instructions (8 rows x 128 B each); a wave issues instructions
  //      wave * L + e; instruction j fills LDS rows r' = 8 j + (lane >> 3) with chunk position lane & 7, i.e. source chunk
  //      (lane & 7) ^ ((r' >> 1) & 7).  x half h holds rows h XH/2 .. + XH/2 - 1 of both wave bands; W half h rows
  //      h WH/4 .. + WH/4 - 1 of the four wave columns (SiLU form: half 0 = gate rows, half 1 = the up rows of the same
  //      output columns). ----
  const T* srcx[2][LX];
  const T* srcw[2][LW];
#pragma unroll
  for (int e = 0; e < LX; ++e) {
    const int rp = 8 * (wave * LX + e) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rp >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int xrow = m0 + XH * (rp / (XH / 2)) + (XH / 2) * h + (rp % (XH / 2));
      if (GR) {
        const int id = grp.sorted_ids[xrow];
        xrow = id < grp.num_valid ? id / grp.top_k_div : 0;      // padding entries read row 0 (discarded)
      } else {
        xrow = min(xrow, M - 1);
      }
      srcx[h][e] = x + (int64_t)xrow * ldx + (int64_t)kt0 * 64 + chunk * 8;
    }
  }
#pragma unroll
  for (int e = 0; e < LW; ++e) {
    const int rp = 8 * (wave * LW + e) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rp >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int wrow, wlim;
      if (EPI == G8_SILU_MUL) {
        wrow = h * (N / 2) + n0 + (WH / 4) * (rp / (WH / 4)) + (rp % (WH / 4));
        wlim = (h + 1) * (N / 2) - 1;
      } else {
        wrow = n0 + (WH / 2) * (rp / (WH / 4)) + (WH / 4) * h + (rp % (WH / 4));
        wlim = N - 1;
      }
      srcw[h][e] = w + (int64_t)min(wrow, wlim) * K + (int64_t)kt0 * 64 + chunk * 8;
    }
  }
  char* const dstx = g8_smem + (wave * LX) * 1024;                        // + parity + half + e KB
  char* const dstw = g8_smem + 2 * G::kXHalf + (wave * LW) * 1024;
  const int last_kt = nkt - 1;
  auto stage_x = [&](int h, int kt) __attribute__((always_inline)) {      // x half h of K step kt -> parity kt & 1
    const int64_t k = (int64_t)min(kt, last_kt) * 64;                     // past the end: re-load the last step (keeps the
    char* dst = dstx + (kt & 1) * G::kParity + h * G::kXHalf;             //   vmcnt arithmetic uniform; nobody reads it)
#pragma unroll
    for (int e = 0; e < LX; ++e) G8_GLDS(srcx[h][e] + k, dst + e * 1024, 0);
  };
  auto stage_w = [&](int h, int kt) __attribute__((always_inline)) {
    const int64_t k = (int64_t)min(kt, last_kt) * 64;
    char* dst = dstw + (kt & 1) * G::kParity + h * G::kWHalf;
#pragma unroll
    for (int e = 0; e < LW; ++e) {
      if (NT) G8_GLDS(srcw[h][e] + k, dst + e * 1024, 2);
      else G8_GLDS(srcw[h][e] + k, dst + e * 1024, 0);
    }
  };

  // ---- fragment read addresses: row 16 i + c16 of this wave's rows in a half-tile, chunk 4 s + q4 at position
  //      ^ ((c16 >> 1) & 7) (the wave's first row is a multiple of 16, so only c16 enters the swizzle) ----
  const uint32_t smem_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)g8_smem;
  const int swz = (c16 >> 1) & 7;
  uint32_t off_s[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) off_s[s] = c16 * 128 + (((4 * s + q4) ^ swz) << 4);
  const uint32_t xa = smem_addr + wr * ((XH / 2) * 128);                    // this wave's XH / 2 rows of an x half-tile
  const uint32_t wa = smem_addr + 2 * G::kXHalf + wc * ((WH / 4) * 128);    // this wave's WH / 4 rows of a W half-tile

  g8_f32x4 acc[2 * MT2][2 * NT2];
#pragma unroll
  for (int i = 0; i < 2 * MT2; ++i)
#pragma unroll
    for (int j = 0; j < 2 * NT2; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
  G8Frag fx[MT2][2], fw[2 * NT2][2];     // x fragments of the current half; W fragments: [0 .. NT2) lo, [NT2 .. 2 NT2) hi

  // ---- prologue: K steps 0 and 1 up to the point the steady state expects (issue order = order of first use) ----
  stage_x(0, 0), stage_w(0, 0), stage_w(1, 0), stage_x(1, 0), stage_x(0, 1), stage_w(0, 1);
  g8_wait_vm<G::kInFlight>();         // x0(0), W0(0) landed (this wave's share)
  g8_barrier();
  if (wr == 1) g8_barrier();          // half a phase behind group 0 from here on

  auto mfma_quadrant = [&](int i0, int j0) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MT2; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j)
          acc[i0 + i][j0 + j] = G8Mfma<T>::mma(fw[j0 + j][s], fx[i][s], acc[i0 + i][j0 + j]);
    __builtin_amdgcn_s_setprio(0);
  };

  for (int kt = 0; kt < nkt; ++kt) {
    const uint32_t par = (kt & 1) * G::kParity;
    // phase 1: x half 0 + W half 0 of this step; DMA W1(kt + 1)
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fw[j][s].u = g8_lds_read16(wa + par + off_s[s], j * 2048);
#pragma unroll
    for (int i = 0; i < MT2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fx[i][s].u = g8_lds_read16(xa + par + off_s[s], i * 2048);
    stage_w(1, kt + 1);
    g8_wait_vm<G::kInFlight>();       // W1(kt) landed; x1(kt), x0(kt+1), W0(kt+1), W1(kt+1) may be in flight
    g8_barrier();
    g8_wait_lgkm0();
    mfma_quadrant(0, 0);
    g8_barrier();
    // phase 2: W half 1; DMA x1(kt + 1)
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fw[NT2 + j][s].u = g8_lds_read16(wa + par + G::kWHalf + off_s[s], j * 2048);
    stage_x(1, kt + 1);
    g8_wait_vm<G::kInFlight>();       // x1(kt) landed
    g8_barrier();
    g8_wait_lgkm0();
    mfma_quadrant(0, NT2);
    g8_barrier();
    // phase 3: x half 1 (into the x registers); DMA x0(kt + 2)
#pragma unroll
    for (int i = 0; i < MT2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fx[i][s].u = g8_lds_read16(xa + par + G::kXHalf + off_s[s], i * 2048);
    stage_x(0, kt + 2);
    g8_barrier();
    g8_wait_lgkm0();
    mfma_quadrant(MT2, NT2);
    g8_barrier();
    // phase 4: nothing to read (W lo is still in registers); DMA W0(kt + 2)
    stage_w(0, kt + 2);
    g8_wait_vm<G::kInFlight>();       // x0(kt+1), W0(kt+1) landed
    g8_barrier();
    mfma_quadrant(MT2, 0);
    g8_barrier();
  }
  g8_wait_vm<0>();                    // no DMA may land in LDS that the next workgroup owns
  if (wr == 0) g8_barrier();          // the barrier group 1 spent up front

  // ---- epilogue: lane holds out[m = m0 + XH wr + 16 i + c16][n = n0 + (WH / 2) wc + 16 j + 4 q4 + r] ----
  const int mb = m0 + XH * wr + c16;
  if (planes != nullptr) {
    float* pl = planes + (int64_t)ks * M * N;
#pragma unroll
    for (int i = 0; i < 2 * MT2; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NT2; ++j) {
        const int m = mb + 16 * i;
        int n, nlim;
        if (EPI == G8_SILU_MUL) {
          n = (j / NT2) * (N / 2) + n0 + (WH / 4) * wc + 16 * (j % NT2) + 4 * q4;
          nlim = ((j / NT2) + 1) * (N / 2);
        } else {
          n = n0 + (WH / 2) * wc + 16 * j + 4 * q4;
          nlim = N;
        }
        if (m < M && n < nlim) *reinterpret_cast<g8_f32x4*>(pl + (int64_t)m * N + n) = acc[i][j];
      }
    return;
  }
  if (EPI == G8_PLAIN) {
#pragma unroll
    for (int i = 0; i < 2 * MT2; ++i) {
      int m = mb + 16 * i;
      float scale = 1.f;
      if (GR) {
        m = grp.sorted_ids[m];                                   // output row = the routed entry itself
        if (m < grp.num_valid && grp.mul_routed_weight) scale = grp.topk_weights[m];
      }
#pragma unroll
      for (int j = 0; j < 2 * NT2; ++j) {
        const int n = n0 + (WH / 2) * wc + 16 * j + 4 * q4;
        if (m < M && n < N) {
          g8_f32x4 v = acc[i][j];
          if (GR) v *= scale;
          uint2 p;
          p.x = (uint32_t)Elem<T>::from_f(v[0]).v | ((uint32_t)Elem<T>::from_f(v[1]).v << 16);
          p.y = (uint32_t)Elem<T>::from_f(v[2]).v | ((uint32_t)Elem<T>::from_f(v[3]).v << 16);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = p;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2 * MT2; ++i) {
      int m = mb + 16 * i;
      if (GR) m = grp.sorted_ids[m];
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const int n = n0 + (WH / 4) * wc + 16 * j + 4 * q4;
        if (m < M && n < n_cols) {
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // the unfused pair rounds the GEMM output to T before the activation reads it
            const float gq = Elem<T>::to_f(Elem<T>::from_f(acc[i][j][e])), uq = Elem<T>::to_f(Elem<T>::from_f(acc[i][NT2 + j][e]));
            r[e] = gq / (1.f + __expf(-gq)) * uq;
            // keep the product an fp32 VALUE: left alone the compiler folds multiply + conversion into one
            // v_fma_mixlo_f16 (a single rounding), one ulp away from silu_and_mul's two in rare cases
            asm volatile("" : "+v"(r[e]));
          }
          uint2 p;
          p.x = (uint32_t)Elem<T>::from_f(r[0]).v | ((uint32_t)Elem<T>::from_f(r[1]).v << 16);
          p.y = (uint32_t)Elem<T>::from_f(r[2]).v | ((uint32_t)Elem<T>::from_f(r[3]).v << 16);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = p;
        }
      }
    }
  }
}

// out[m, n] = T(sum_z planes[z][m][n]) in slice order (SiLU * mul variant: gate / up columns n, n + N / 2)
template <typename T, int EPI>
__global__ void __launch_bounds__(256)
gemm8p_reduce_kernel(T* __restrict__ out, const float* __restrict__ planes, int ksplit, int M, int N, int64_t ldo) {
  const int n_out = EPI == G8_SILU_MUL ? N / 2 : N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = n_out / 4;
  if (i >= (int64_t)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
  const int64_t plane = (int64_t)M * N;
  const float* srcp = planes + (int64_t)m * N + n;
  g8_f32x4 a = *reinterpret_cast<const g8_f32x4*>(srcp);
  for (int z = 1; z < ksplit; ++z) a += *reinterpret_cast<const g8_f32x4*>(srcp + z * plane);
  float r[4] = {a[0], a[1], a[2], a[3]};
  if (EPI == G8_SILU_MUL) {
    g8_f32x4 u = *reinterpret_cast<const g8_f32x4*>(srcp + N / 2);
    for (int z = 1; z < ksplit; ++z) u += *reinterpret_cast<const g8_f32x4*>(srcp + N / 2 + z * plane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gq = Elem<T>::to_f(Elem<T>::from_f(r[j])), uq = Elem<T>::to_f(Elem<T>::from_f(u[j]));
      r[j] = gq / (1.f + __expf(-gq)) * uq;
      asm volatile("" : "+v"(r[j]));   // see the kernel's epilogue
    }
  }
  uint2 p;
  p.x = (uint32_t)Elem<T>::from_f(r[0]).v | ((uint32_t)Elem<T>::from_f(r[1]).v << 16);
  p.y = (uint32_t)Elem<T>::from_f(r[2]).v | ((uint32_t)Elem<T>::from_f(r[3]).v << 16);
  *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = p;
}

static std::atomic<int> g_g8_cus{256};

// K slices: the tiles x slices workgroups run in whole rounds of the share's CUs (one workgroup per CU); pick the
// split with the least (rounds x K steps per slice), charging each extra slice its fp32 plane round trip.
static int g8_pick_ksplit(int tiles, int nkt, int M, int N, size_t planes_bytes) {
  const int cus = max(8, g_g8_cus.load(std::memory_order_relaxed));
  int best = 1;
  float best_cost = 1e30f;
  for (int ksp = 1; ksp <= 8; ++ksp) {
    const int per = (nkt + ksp - 1) / ksp;
    if (ksp > 1 && (per < 8 || (size_t)ksp * M * N * 4 > planes_bytes)) break;
    const int eff = (nkt + per - 1) / per;
    if (eff != ksp) continue;
    const int rounds = (tiles * eff + cus - 1) / cus;
    // a K step of a tile ~0.9 us of matrix work; a plane costs M x N x 8 bytes of traffic spread over the share
    float cost = rounds * (per * 0.9f + 3.f);
    if (eff > 1) cost += eff * (8.f * M * N / (cus * 30e3f)) + 3.f;
    if (cost < best_cost - 1e-3f) best_cost = cost, best = eff;
  }
  return best;
}

template <typename T, int EPI, int XH>
static int g8_launch_geo(T* out, float* planes, size_t planes_bytes, const T* x, const T* w, int M, int N, int K, int64_t ldx,
                         int64_t ldo, int force_ks, hipStream_t st, int* planes_only_ks = nullptr) {
  using G = G8Geo<XH>;
  const int n_cols = EPI == G8_SILU_MUL ? N / 2 : N;
  const int cols_per_tile = EPI == G8_SILU_MUL ? G::WH : G::BN;
  const int tiles_n = (n_cols + cols_per_tile - 1) / cols_per_tile, tiles_m = (M + G::BM - 1) / G::BM;
  const int tiles = tiles_n * tiles_m;
  const int nkt = K / 64;
  int ksp = 1;
  if (planes) ksp = force_ks > 0 ? min(force_ks, nkt) : g8_pick_ksplit(tiles, nkt, M, N, planes_bytes);
  while (ksp > 1 && (size_t)ksp * M * N * 4 > planes_bytes) --ksp;
  const int per = (nkt + ksp - 1) / ksp;
  ksp = (nkt + per - 1) / per;
  static std::atomic<uint64_t> lds_ok_nt{0}, lds_ok{0};
  if (tiles_m == 1) {
    if (ensure_dynamic_lds((const void*)gemm8p_kernel<T, EPI, true, XH>, G::kLds, lds_ok_nt, "gemm8p")) return 1;
    hipLaunchKernelGGL((gemm8p_kernel<T, EPI, true, XH>), dim3(tiles, ksp), dim3(512), G::kLds, st, out,
                       ksp > 1 ? planes : (float*)nullptr, x, w, M, N, K, ldx, ldo, per);
  } else {
    // XCD-contiguous order pays where the weight does not fit the caches between row tiles (vocabulary-sized heads:
    // 128 256 x 4096 at 1024 rows 1073 -> 834 us) and is neutral to slightly negative for layer-sized weights
    // (profiles/r03_kbench_gemm_tall_tile_order.txt); SEMIPD_G8_XCD_ORDER=0 / 1 forces it
    static const int xcd_knob = []() { const char* e = getenv("SEMIPD_G8_XCD_ORDER"); return e ? atoi(e) : -1; }();
    const bool xcd_order = xcd_knob >= 0 ? xcd_knob != 0 : n_cols >= 32768;
    static std::atomic<uint64_t> lds_ok_x{0};
    if (xcd_order) {
      if (ensure_dynamic_lds((const void*)gemm8p_kernel<T, EPI, false, XH, false, 1>, G::kLds, lds_ok_x, "gemm8p")) return 1;
      hipLaunchKernelGGL((gemm8p_kernel<T, EPI, false, XH, false, 1>), dim3(tiles, ksp), dim3(512), G::kLds, st, out,
                         ksp > 1 ? planes : (float*)nullptr, x, w, M, N, K, ldx, ldo, per);
    } else {
      if (ensure_dynamic_lds((const void*)gemm8p_kernel<T, EPI, false, XH>, G::kLds, lds_ok, "gemm8p")) return 1;
      hipLaunchKernelGGL((gemm8p_kernel<T, EPI, false, XH>), dim3(tiles, ksp), dim3(512), G::kLds, st, out,
                         ksp > 1 ? planes : (float*)nullptr, x, w, M, N, K, ldx, ldo, per);
    }
  }
  int rc = launch_status("gemm8p");
  if (planes_only_ks) {   // the consumer sums the K-slice planes (semipd_fused_add_rmsnorm_planes); one slice: `out` is written
    *planes_only_ks = ksp;
    return rc;
  }
  if (rc || ksp == 1) return rc;
  const int64_t items = (int64_t)M * (n_cols / 4);
  hipLaunchKernelGGL((gemm8p_reduce_kernel<T, EPI>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, out,
                     (const float*)planes, ksp, M, N, ldo);
  return launch_status("gemm8p_reduce");
}

template <typename T, int EPI>
static int g8_launch(T* out, float* planes, size_t planes_bytes, const T* x, const T* w, int M, int N, int K, int64_t ldx,
                     int64_t ldo, int force_ks, int force_geo, hipStream_t st, int* planes_only_ks = nullptr) {
  // up to 128 rows: the 128 x 512 tile (no matrix work on padding rows, twice the weight rows per MFMA)
  const bool narrow = force_geo ? force_geo == 64 : M <= 128;
  if (narrow) return g8_launch_geo<T, EPI, 64>(out, planes, planes_bytes, x, w, M, N, K, ldx, ldo, force_ks, st, planes_only_ks);
  return g8_launch_geo<T, EPI, 128>(out, planes, planes_bytes, x, w, M, N, K, ldx, ldo, force_ks, st, planes_only_ks);
}

template <typename T, int EPI, int XH>
static int g8_launch_grouped(T* c, const T* a, const T* w, const G8Group& grp, int64_t max_sorted, int N, int K, int64_t lda,
                             int64_t ldc, hipStream_t st) {
  using G = G8Geo<XH>;
  const int n_cols = EPI == G8_SILU_MUL ? N / 2 : N;
  const int cols_per_tile = EPI == G8_SILU_MUL ? G::WH : G::BN;
  const int tiles_n = (n_cols + cols_per_tile - 1) / cols_per_tile, tiles_m = (int)(max_sorted / G::BM);
  if (tiles_m == 0) return 0;
  static std::atomic<uint64_t> lds_ok{0};
  if (ensure_dynamic_lds((const void*)gemm8p_kernel<T, EPI, false, XH, true>, G::kLds, lds_ok, "gemm8p_grouped")) return 1;
  hipLaunchKernelGGL((gemm8p_kernel<T, EPI, false, XH, true>), dim3(tiles_m * tiles_n, 1), dim3(512), G::kLds, st, c,
                     (float*)nullptr, a, w, grp.num_valid, N, K, lda, ldc, K / 64, grp);
  return launch_status("gemm8p_grouped");
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_gemm_tall_set_cus(int cus) {
  SEMIPD_CHECK_ARG(cus >= 0 && cus <= 4096, SEMIPD_EINVAL, "gemm_tall_set_cus: bad CU count %d", cus);
  g_g8_cus.store(cus == 0 ? 256 : cus, std::memory_order_relaxed);
  owned_cus().store(cus, std::memory_order_relaxed);
  return 0;
}

int semipd_gemm_tall(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes, int64_t rows,
                     int64_t n, int64_t k, int64_t ldx, int64_t ldo, int fuse_silu_mul, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(rows >= 0 && n > 0 && k > 0 && ldx >= k, SEMIPD_EINVAL, "gemm_tall: bad sizes");
  if (rows == 0) return 0;
  SEMIPD_CHECK_ARG(out && x && weight, SEMIPD_EINVAL, "gemm_tall: null pointer");
  const int64_t n_out = fuse_silu_mul ? n / 2 : n;
  SEMIPD_CHECK_ARG(ldo >= n_out && (!fuse_silu_mul || n % 2 == 0), SEMIPD_EINVAL, "gemm_tall: ldo < output width");
  SEMIPD_CHECK_ARG(k % 64 == 0 && ldx % 8 == 0 && n_out % 16 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(weight) &&
                       (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && n < (1 << 30) && k < (1 << 30) && rows < (1 << 30) &&
                       (!workspace || aligned16(workspace)),
                   SEMIPD_EALIGN, "gemm_tall: k %% 64, output width %% 16, 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  const char* e = getenv("SEMIPD_G8_KS");
  const int force_ks = e ? atoi(e) : 0;
  const char* eg = getenv("SEMIPD_G8_XH");     // kbench / tests: force the 256 x 256 (128) or the 128 x 512 (64) tile
  const int force_geo = eg ? atoi(eg) : 0;
  int rc = 0;
  if (fuse_silu_mul) {
    SEMIPD_DISPATCH_HALF(dtype, T, rc = (g8_launch<T, G8_SILU_MUL>((T*)out, (float*)workspace, workspace_bytes, (const T*)x,
                                                                   (const T*)weight, (int)rows, (int)n, (int)k, ldx, ldo,
                                                                   force_ks, force_geo, st)));
  } else {
    SEMIPD_DISPATCH_HALF(dtype, T, rc = (g8_launch<T, G8_PLAIN>((T*)out, (float*)workspace, workspace_bytes, (const T*)x,
                                                                (const T*)weight, (int)rows, (int)n, (int)k, ldx, ldo, force_ks,
                                                                force_geo, st)));
  }
  return rc;
}

/* The same GEMM (plain epilogue), stopped before the reduction over its K slices: the split is chosen as semipd_gemm_tall
 * chooses it; *ksplit > 1: fp32 planes [*ksplit][rows][n] in `planes` and `out` untouched -- the consumer sums them in slice
 * order and rounds to dtype (semipd_fused_add_rmsnorm_planes: the bits of semipd_gemm_tall followed by the fused add + norm,
 * one launch and one round trip of the [rows, n] result fewer); *ksplit == 1: `out` holds the result.  The row-parallel layers
 * of a prefill batch (o_proj / down_proj -> RMSNorm(x, residual), models/llama.py:279-302) where the tiled GEMM is preferred. */
int semipd_gemm_tall_planes(void* out, float* planes, size_t planes_bytes, const void* x, const void* weight, int64_t rows,
                            int64_t n, int64_t k, int64_t ldx, int64_t ldo, int dtype, int* ksplit, void* stream) {
  SEMIPD_CHECK_ARG(rows > 0 && n > 0 && k > 0 && ldx >= k && ldo >= n && ksplit, SEMIPD_EINVAL, "gemm_tall_planes: bad sizes");
  SEMIPD_CHECK_ARG(out && planes && x && weight, SEMIPD_EINVAL, "gemm_tall_planes: null pointer");
  SEMIPD_CHECK_ARG(k % 64 == 0 && ldx % 8 == 0 && n % 16 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(weight) &&
                       aligned16(planes) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && n < (1 << 30) && k < (1 << 30) &&
                       rows < (1 << 30),
                   SEMIPD_EALIGN, "gemm_tall_planes: k %% 64, n %% 16, 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  const char* e = getenv("SEMIPD_G8_KS");
  const int force_ks = e ? atoi(e) : 0;
  const char* eg = getenv("SEMIPD_G8_XH");
  const int force_geo = eg ? atoi(eg) : 0;
  int rc = 0;
  SEMIPD_DISPATCH_HALF(dtype, T, rc = (g8_launch<T, G8_PLAIN>((T*)out, planes, planes_bytes, (const T*)x, (const T*)weight,
                                                              (int)rows, (int)n, (int)k, ldx, ldo, force_ks, force_geo, st,
                                                              ksplit)));
  return rc;
}

/* invoke_fused_moe_kernel (fused_moe.py:501-612) for prefill-sized calls with the tiled ping-pong GEMM: sorted_token_ids /
 * expert_ids from moe_align_block_size with block size block_m = 256 (256 x 256 tiles) or 128 (128 x 512 tiles): one
 * expert per tile; sorted_token_ids must hold max_sorted entries, a multiple of block_m.  c[id, :] = a[id / top_k_div, :] @ w[expert]^T for every routed entry id <
 * num_valid, times topk_weights[id] when mul_routed_weight; fuse_silu_mul: w[e] = merged [gate; up] ([n, k], n = 2 x
 * output width) and c = SiLU(gate) * up of the products rounded to dtype. */
int semipd_moe_gemm_tall(void* c, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
                         const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n, int64_t k,
                         int64_t max_sorted, int top_k_div, int mul_routed_weight, int fuse_silu_mul, int block_m, int dtype,
                         void* stream) {
  SEMIPD_CHECK_ARG(num_valid >= 0 && n > 0 && k > 0 && max_sorted >= 0 && top_k_div > 0, SEMIPD_EINVAL, "moe_gemm_tall: bad sizes");
  if (num_valid == 0 || max_sorted == 0) return 0;
  SEMIPD_CHECK_ARG(c && a && w && sorted_token_ids && expert_ids && num_tokens_post_pad, SEMIPD_EINVAL,
                   "moe_gemm_tall: null pointer");
  SEMIPD_CHECK_ARG(!mul_routed_weight || topk_weights, SEMIPD_EINVAL, "moe_gemm_tall: topk_weights required");
  const int64_t n_out = fuse_silu_mul ? n / 2 : n;
  SEMIPD_CHECK_ARG((block_m == 256 || block_m == 128) && max_sorted % block_m == 0 && k % 64 == 0 && n_out % 16 == 0 && (!fuse_silu_mul || n % 2 == 0) && aligned16(a) &&
                       aligned16(w) && (reinterpret_cast<uintptr_t>(c) & 7u) == 0 && num_valid < (1 << 30) && n < (1 << 30) &&
                       k < (1 << 30),
                   SEMIPD_ESHAPE, "moe_gemm_tall: block size 256 or 128, k %% 64, output width %% 16 required");
  G8Group grp{sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights, (int)num_valid, top_k_div, mul_routed_weight};
  int rc = 0;
#define G8G(EPIV, XHV) \
  SEMIPD_DISPATCH_HALF(dtype, T, rc = (g8_launch_grouped<T, EPIV, XHV>((T*)c, (const T*)a, (const T*)w, grp, max_sorted, (int)n, (int)k, k, n_out, as_stream(stream))))
  if (block_m == 256) {
    if (fuse_silu_mul) { G8G(G8_SILU_MUL, 128); } else { G8G(G8_PLAIN, 128); }
  } else {
    if (fuse_silu_mul) { G8G(G8_SILU_MUL, 64); } else { G8G(G8_PLAIN, 64); }
  }
#undef G8G
  return rc;
}

}  // extern "C"
