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

// ---------------------------------------------------------------------------------------------------------------------
// The same 256 x 256 x 64 tile on FOUR waves (round 6): one wave per SIMD, each with a 128 x 128 quarter of the tile --
// 8 x 8 accumulator tiles = 256 registers, which the compiler keeps in the accumulation half of the unified register file.
// Why: the 8-wave form reads (128 + 64) rows x 128 B of fragments per wave and K step, 192 KB per CU, and its LDS-DMA writes
// another 64 KB: 2048 cycles of the LDS's 128 B / cycle -- exactly the 2048 cycles the K step's MFMAs take.  The kernel is
// bound by LDS bandwidth as much as by the matrix pipe (1.5-1.66 us per K step measured where the pipe needs 1.37 at the
// ~1.5 GHz the chip sustains under it).  A 128 x 128 wave tile needs (128 + 128) rows per K step: 128 KB per CU + 64 KB of
// DMA = 1536 cycles.
//   * LDS image, DMA and swizzle as above (half-tiles of 128 rows, 2 parities x 4 half-tiles = 128 KB), except that the W
//     half-tiles hold rows 0-63 / 64-127 of each of TWO 128-row wave columns;
//   * a K step is four phases of 32 MFMAs (one 64 x 64 quadrant, both k halves).  There is no partner wave on the SIMD to
//     cover the fragment reads, so the wave covers them itself: all four operand halves (x0, x1, W0, W1: 4 x 32 registers)
//     live in registers, every phase reads ONE half for a LATER phase while its own MFMAs run, and the phase order
//     alternates between K steps so that exactly one half falls free per phase:
//         even step  (x0,W0) (x0,W1) (x1,W1) (x1,W0)     reads  W1(k) x1(k) x0(k+1) W1(k+1)
//         odd step   (x0,W1) (x0,W0) (x1,W0) (x1,W1)     reads  W0(k) x1(k) x0(k+1) W0(k+1)
//   * a half-tile's LDS slot is re-filled (for K step + 2) in the phase after it was read and is read 8 phases after that:
//     7 phases (~2 us of matrix work) of flight, six younger half-tiles behind every wait: s_waitcnt vmcnt(24);
//   * phase = { vmcnt(24): my share of the half to read now; lgkmcnt(0): last phase's reads; s_barrier; 8 ds_read_b128;
//     4 DMA; 32 MFMAs }.
// The MFMAs of the 4-wave kernel as inline asm with the accumulator tile tied to an ACCUMULATION register ("+a"): 256 of the
// wave's 512 registers are accumulators, and left to itself the register allocator moves fragments into the accumulation half
// too, spills accumulators and copies them around (1.2 KB of scratch per lane, ~370 v_accvgpr moves inside the K loop in the
// first build).  With the classes fixed -- accumulators in a[0:255], fragments and addresses in v[0:255] -- nothing spills.
// asm volatile also keeps the issue order of a phase: the fragment reads and the DMA first, then the 32 MFMAs.
template <typename T> struct G4Mfma;
template <> struct G4Mfma<bf16_t> {
  __device__ static inline void mma(g8_f32x4& c, const G8Frag& a, const G8Frag& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a.b), "v"(b.b));
  }
};
template <> struct G4Mfma<f16_t> {
  __device__ static inline void mma(g8_f32x4& c, const G8Frag& a, const G8Frag& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a.f), "v"(b.f));
  }
};

template <typename T, int EPI, bool NT, bool GR = false, int ORDER = 0>
__global__ void __launch_bounds__(256)
gemm4w_kernel(T* __restrict__ out, float* __restrict__ planes, const T* __restrict__ x, const T* __restrict__ w, int M, int N,
              int K, int64_t ldx, int64_t ldo, int kt_per_slice, G8Group grp = G8Group()) {
  constexpr int kHalf = 128 * 128;            // a half-tile: 128 rows x 128 B
  constexpr int kParity = 4 * kHalf;          // x0 x1 W0 W1 of one K-step parity
  constexpr int L = 4;                        // DMA instructions per wave and half-tile (16 pieces of 8 rows, 4 waves)
  extern __shared__ __attribute__((aligned(16))) char g8_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int n_cols = EPI == G8_SILU_MUL ? N / 2 : N;
  constexpr int cols_per_tile = EPI == G8_SILU_MUL ? 128 : 256;
  const int tiles_n = (n_cols + cols_per_tile - 1) / cols_per_tile;
  int tile_m, tile_n;
  if (GR || ORDER == 0) {
    tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  } else {
    const int nwg = gridDim.x, id = blockIdx.x, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    const int tiles_m = nwg / tiles_n;
    tile_n = logical / tiles_m, tile_m = logical - tile_n * tiles_m;
  }
  const int m0 = tile_m * 256, n0 = tile_n * cols_per_tile;
  const int ks = blockIdx.y;
  const int nkt_total = K >> 6;
  const int kt0 = ks * kt_per_slice;
  const int nkt = min(kt_per_slice, nkt_total - kt0);
  if (GR) {
    if (m0 >= grp.num_post_pad[0]) return;
    w += (int64_t)grp.expert_ids[tile_m] * N * K;
  }

  // ---- DMA sources: piece p = wave * L + e fills LDS rows 8 p + (lane >> 3) of a half-tile ----
  const T* srcx[2][L];
  const T* srcw[2][L];
#pragma unroll
  for (int e = 0; e < L; ++e) {
    const int rp = 8 * (wave * L + e) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rp >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int xrow = m0 + 128 * (rp >> 6) + 64 * h + (rp & 63);       // rows 64 h .. + 63 of both 128-row wave bands
      if (GR) {
        const int id = grp.sorted_ids[xrow];
        xrow = id < grp.num_valid ? id / grp.top_k_div : 0;
      } else {
        xrow = min(xrow, M - 1);
      }
      srcx[h][e] = x + (int64_t)xrow * ldx + (int64_t)kt0 * 64 + chunk * 8;
      int wrow, wlim;
      if (EPI == G8_SILU_MUL) {
        wrow = h * (N / 2) + n0 + 64 * (rp >> 6) + (rp & 63);     // half 0 = gate rows, half 1 = the up rows of the same columns
        wlim = (h + 1) * (N / 2) - 1;
      } else {
        wrow = n0 + 128 * (rp >> 6) + 64 * h + (rp & 63);         // rows 64 h .. + 63 of both 128-row wave columns
        wlim = N - 1;
      }
      srcw[h][e] = w + (int64_t)min(wrow, wlim) * K + (int64_t)kt0 * 64 + chunk * 8;
    }
  }
  char* const dst0 = g8_smem + (wave * L) * 1024;
  const int last_kt = nkt - 1;
  // half id: 0 = x0, 1 = x1, 2 = W0, 3 = W1 (the order of the slots inside a parity)
  auto stage = [&](int half, int kt) __attribute__((always_inline)) {
    const int64_t k = (int64_t)min(kt, last_kt) * 64;               // past the end: the last step again (nobody multiplies it)
    char* dst = dst0 + (kt & 1) * kParity + half * kHalf;
#pragma unroll
    for (int e = 0; e < L; ++e) {
      if (half < 2) G8_GLDS(srcx[half][e] + k, dst + e * 1024, 0);
      else if (NT) G8_GLDS(srcw[half - 2][e] + k, dst + e * 1024, 2);
      else G8_GLDS(srcw[half - 2][e] + k, dst + e * 1024, 0);
    }
  };

  const uint32_t smem_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)g8_smem;
  const int swz = (c16 >> 1) & 7;
  uint32_t off_s[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) off_s[s2] = c16 * 128 + (((4 * s2 + q4) ^ swz) << 4);
  const uint32_t xa = smem_addr + wr * (64 * 128);                  // this wave's 64 rows of an x half-tile
  const uint32_t wa = smem_addr + 2 * kHalf + wc * (64 * 128);      // this wave's 64 rows of a W half-tile

  g8_f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
  G8Frag fx0[4][2], fx1[4][2], fw0[4][2], fw1[4][2];

#define G4_READ(F, BASE)                                                              \
  _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) F[t_][s_].u = g8_lds_read16((BASE) + off_s[s_], t_ * 2048)
#define G4_SYNC()                 \
  g8_wait_vm<24>();               \
  g8_wait_lgkm0();                \
  g8_barrier()
  // One phase: 32 MFMAs (quadrant I0, J0 from the fragments XF, WF) with the reads of half-tile RF (8 ds_read_b128) and the
  // DMA of one half-tile (4 instructions) issued BETWEEN them -- the matrix pipe starts right behind the barrier instead of
  // after ~100 cycles of issue, and nothing of this phase's MFMAs depends on what is read or staged here.
#define G4_PHASE(RF, RBASE, SHALF, SKT, XF, WF, I0, J0)                                                             \
  do {                                                                                                             \
    G4_SYNC();                                                                                                     \
    const int64_t k_ = (int64_t)min((SKT), last_kt) * 64;                                                          \
    char* const d_ = dst0 + ((SKT) & 1) * kParity + (SHALF) * kHalf;                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                           \
          G4Mfma<T>::mma(acc[(I0) + i_][(J0) + j_], WF[j_][s_], XF[i_][s_]);                                       \
        if (s_ == 0) {                                                                                             \
          RF[i_][0].u = g8_lds_read16((RBASE) + off_s[0], i_ * 2048);                                              \
          RF[i_][1].u = g8_lds_read16((RBASE) + off_s[1], i_ * 2048);                                              \
        } else if ((SHALF) < 2) {                                                                                  \
          G8_GLDS(srcx[(SHALF) & 1][i_] + k_, d_ + i_ * 1024, 0);                                                  \
        } else if (NT) {                                                                                           \
          G8_GLDS(srcw[(SHALF) & 1][i_] + k_, d_ + i_ * 1024, 2);                                                  \
        } else {                                                                                                   \
          G8_GLDS(srcw[(SHALF) & 1][i_] + k_, d_ + i_ * 1024, 0);                                                  \
        }                                                                                                          \
      }                                                                                                            \
  } while (0)

  // ---- prologue: the first eight half-tiles in the order they are read ----
  stage(0, 0), stage(2, 0), stage(3, 0), stage(1, 0), stage(0, 1), stage(3, 1), stage(2, 1), stage(1, 1);
  g8_wait_vm<28>();
  g8_barrier();
  G4_READ(fx0, xa);                                   // x0(0)
  G4_SYNC();
  G4_READ(fw0, wa);                                   // W0(0)
  stage(0, 2);                                        // x0(0) has been read by everybody: its slot takes x0(2)

  for (int kt = 0; kt < nkt; kt += 2) {
    const uint32_t pa = (kt & 1) * kParity, pb = pa ^ kParity;     // parity of this step | of the next one
    // ---- even step: (x0,W0) (x0,W1) (x1,W1) (x1,W0); each phase reads one half-tile for a later phase and re-fills the
    //      slot of the half-tile read one phase ago (K step + 2) ----
    G4_PHASE(fw1, wa + pa + kHalf, 2, kt + 2, fx0, fw0, 0, 0);      // reads W1(kt);     W0(kt) -> W0(kt + 2)
    G4_PHASE(fx1, xa + pa + kHalf, 3, kt + 2, fx0, fw1, 0, 4);      // reads x1(kt);     W1(kt) -> W1(kt + 2)
    G4_PHASE(fx0, xa + pb, 1, kt + 2, fx1, fw1, 4, 4);              // reads x0(kt + 1); x1(kt) -> x1(kt + 2)
    G4_PHASE(fw1, wa + pb + kHalf, 0, kt + 3, fx1, fw0, 4, 0);      // reads W1(kt + 1); x0(kt + 1) -> x0(kt + 3)
    // ---- odd step: (x0,W1) (x0,W0) (x1,W0) (x1,W1).  A slice is an EVEN count of K steps (the launcher sees to it: one loop
    //      exit and no branch around the MFMAs -- with either, the register allocator parks accumulators in scratch) ----
    G4_PHASE(fw0, wa + pb, 3, kt + 3, fx0, fw1, 0, 4);              // reads W0(kt + 1); W1(kt + 1) -> W1(kt + 3)
    G4_PHASE(fx1, xa + pb + kHalf, 2, kt + 3, fx0, fw0, 0, 0);      // reads x1(kt + 1); W0(kt + 1) -> W0(kt + 3)
    G4_PHASE(fx0, xa + pa, 1, kt + 3, fx1, fw0, 4, 0);              // reads x0(kt + 2); x1(kt + 1) -> x1(kt + 3)
    G4_PHASE(fw0, wa + pa, 0, kt + 4, fx1, fw1, 4, 4);              // reads W0(kt + 2); x0(kt + 2) -> x0(kt + 4)
  }
#undef G4_READ
#undef G4_PHASE
#undef G4_SYNC
  g8_wait_vm<0>();                    // no DMA may land in LDS that the next workgroup owns
  g8_wait_lgkm0();
  // the last MFMAs retire before their accumulators are read: the compiler knows nothing of the hazards of an asm MFMA, and
  // only an asm that DEFINES the registers keeps its v_accvgpr_reads behind it (the first build read acc[7][7] one k-half short)
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(acc[7][4]), "+a"(acc[7][5]), "+a"(acc[7][6]), "+a"(acc[7][7]));

  // ---- epilogue: lane holds out[m = m0 + 128 wr + 16 i + c16][n = n0 + 128 wc + 16 j + 4 q4 + r] ----
  const int mb = m0 + 128 * wr + c16;
  if (planes != nullptr) {
    float* pl = planes + (int64_t)ks * M * N;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int m = mb + 16 * i;
        int n, nlim;
        if (EPI == G8_SILU_MUL) {
          n = (j / 4) * (N / 2) + n0 + 64 * wc + 16 * (j % 4) + 4 * q4;
          nlim = ((j / 4) + 1) * (N / 2);
        } else {
          n = n0 + 128 * wc + 16 * j + 4 * q4;
          nlim = N;
        }
        if (m < M && n < nlim) *reinterpret_cast<g8_f32x4*>(pl + (int64_t)m * N + n) = acc[i][j];
      }
    return;
  }
  if (EPI == G8_PLAIN) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m = mb + 16 * i;
      float scale = 1.f;
      if (GR) {
        m = grp.sorted_ids[m];
        if (m < grp.num_valid && grp.mul_routed_weight) scale = grp.topk_weights[m];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = n0 + 128 * wc + 16 * j + 4 * q4;
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
    for (int i = 0; i < 8; ++i) {
      int m = mb + 16 * i;
      if (GR) m = grp.sorted_ids[m];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + 64 * wc + 16 * j + 4 * q4;
        if (m < M && n < n_cols) {
          float r[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // the unfused pair rounds the GEMM output to T before the activation reads it
            const float gq = Elem<T>::to_f(Elem<T>::from_f(acc[i][j][e])), uq = Elem<T>::to_f(Elem<T>::from_f(acc[i][4 + j][e]));
            r[e] = gq / (1.f + __expf(-gq)) * uq;
            asm volatile("" : "+v"(r[e]));   // (see gemm8p_kernel's epilogue: two roundings, not one fused one)
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

// Which kernel runs the 256 x 256 tiles: 0 (default) = by epilogue -- four waves (gemm4w_kernel) for the plain and the
// planes epilogue, where it is 2-8 % ahead on every Llama-3-8B layer shape, eight waves for SiLU * mul, where the two tie
// (profiles/r06_kbench_gemm_forms_*.txt); 4 / 8 force one (SEMIPD_G8_FORM, semipd_gemm_tall_set_form)
static std::atomic<int> g_g8_form{-1};
static bool g8_four_waves(int epi) {
  int f = g_g8_form.load(std::memory_order_relaxed);
  if (f < 0) {
    const char* e = getenv("SEMIPD_G8_FORM");
    f = e ? atoi(e) : 0;
    g_g8_form.store(f, std::memory_order_relaxed);
  }
  return f == 4 || (f == 0 && epi == G8_PLAIN);
}

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
  // 256 x 256 tiles: the 8-wave ping-pong kernel or the 4-wave one (one 128 x 128 quarter per wave; see gemm4w_kernel), which
  // walks K two steps at a time: only where every slice is an even count of K steps.  The K partition itself never depends on
  // the form (the SiLU epilogue on eight waves and the plain one on four must sum the same slices: fused == unfused pair)
  const bool four = XH == 128 && g8_four_waves(EPI) && nkt % 2 == 0 && per % 2 == 0;
  static std::atomic<uint64_t> lds_ok_nt{0}, lds_ok{0};
#define G8_GO(K8, K4, OK)                                                                                              \
  do {                                                                                                                 \
    if (four) {                                                                                                        \
      static std::atomic<uint64_t> ok4{0};                                                                             \
      if (ensure_dynamic_lds((const void*)K4, G::kLds, ok4, "gemm4w")) return 1;                                      \
      hipLaunchKernelGGL(K4, dim3(tiles, ksp), dim3(256), G::kLds, st, out, ksp > 1 ? planes : (float*)nullptr, x, w, M, N, \
                         K, ldx, ldo, per);                                                                            \
    } else {                                                                                                           \
      if (ensure_dynamic_lds((const void*)K8, G::kLds, OK, "gemm8p")) return 1;                                       \
      hipLaunchKernelGGL(K8, dim3(tiles, ksp), dim3(512), G::kLds, st, out, ksp > 1 ? planes : (float*)nullptr, x, w, M, N, \
                         K, ldx, ldo, per);                                                                            \
    }                                                                                                                  \
  } while (0)
  if (tiles_m == 1) {
    G8_GO((gemm8p_kernel<T, EPI, true, XH>), (gemm4w_kernel<T, EPI, true>), lds_ok_nt);
  } else {
    // XCD-contiguous order pays where the weight does not fit the caches between row tiles (vocabulary-sized heads:
    // 128 256 x 4096 at 1024 rows 1073 -> 834 us) and is neutral to slightly negative for layer-sized weights
    // (profiles/r03_kbench_gemm_tall_tile_order.txt); SEMIPD_G8_XCD_ORDER=0 / 1 forces it
    static const int xcd_knob = []() { const char* e = getenv("SEMIPD_G8_XCD_ORDER"); return e ? atoi(e) : -1; }();
    const bool xcd_order = xcd_knob >= 0 ? xcd_knob != 0 : n_cols >= 32768;
    static std::atomic<uint64_t> lds_ok_x{0};
    if (xcd_order) {
      G8_GO((gemm8p_kernel<T, EPI, false, XH, false, 1>), (gemm4w_kernel<T, EPI, false, false, 1>), lds_ok_x);
    } else {
      G8_GO((gemm8p_kernel<T, EPI, false, XH>), (gemm4w_kernel<T, EPI, false>), lds_ok);
    }
  }
#undef G8_GO
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
  if (XH == 128 && g8_four_waves(-1) && (K / 64) % 2 == 0) {   // (grouped tiles: 22-32 K steps, the longer prologue costs more than the loop gains: only when forced)
    static std::atomic<uint64_t> ok4{0};
    if (ensure_dynamic_lds((const void*)gemm4w_kernel<T, EPI, false, true>, G::kLds, ok4, "gemm4w_grouped")) return 1;
    hipLaunchKernelGGL((gemm4w_kernel<T, EPI, false, true>), dim3(tiles_m * tiles_n, 1), dim3(256), G::kLds, st, c,
                       (float*)nullptr, a, w, grp.num_valid, N, K, lda, ldc, K / 64, grp);
    return launch_status("gemm4w_grouped");
  }
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

/* 8 = the 8-wave ping-pong kernel for 256 x 256 tiles, 4 = the 4-wave one (gemm4w_kernel), 0 = by epilogue (the default) */
int semipd_gemm_tall_set_form(int waves) {
  SEMIPD_CHECK_ARG(waves == 0 || waves == 4 || waves == 8, SEMIPD_EINVAL, "gemm_tall_set_form: 0 (by epilogue), 4 or 8, got %d", waves);
  g_g8_form.store(waves, std::memory_order_relaxed);
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
