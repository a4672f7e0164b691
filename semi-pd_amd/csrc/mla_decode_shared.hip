// MLA (DeepSeek) paged decode attention, stage 1, for 64 / 128 heads per rank on gfx950 (SURVEY a5 + a13; round 2,
// VERDICT item 6).
//
// With 128 heads on one latent row (576 bf16 = [512 compressed | 64 rope], keys = the whole row, values = its first
// 512 columns) decode is no longer a memory-bound walk: 278 kflop per 1152-byte row.  mla_decode_wide_kernel
// (mla_decode_attention.hip) shares a latent tile between 64 heads, computes S twice per head tile and stages every
// tile through registers and two __syncthreads; it reaches 1.4-1.9 TB/s of rows.  This kernel is built like the
// shared-KV prefill kernel (extend_attention_shared_kv.hip):
//
//   workgroup = (request, group of 32 * NWV heads, kv split), NWV = 4 (or 2) waves, ONE wave per SIMD;
//     a wave owns 32 heads and ALL 512 output dims: O^T is 16 accumulator tiles of 32x32 (256 registers -- the
//     accumulation half of the register file), so the latent tile is read from HBM once per 128 heads and S is
//     computed once.  Q^T is 36 B-operand fragments (144 registers): 24 stay in registers, every third one lives in
//     the wave's own 12 KiB of LDS (all 36 in registers spill: the MFMAs of a 512-register kernel write accumulation
//     registers only, S^T needs 16 of them, and the vector half then has to hold a displaced tile of O^T as well).
//   latent tiles of 32 rows arrive by LDS-DMA (global_load ... lds through kv_indices) into a three-deep ring (3 x 36
//     KiB): a tile is four 256-byte "planes" (dims 0-127, 128-255, ...) of 32 rows + one 128-byte rope plane; a DMA
//     instruction fills 1 KiB = 4 rows of a plane (8 rows of the rope plane).  The tile requested during S^T of
//     tile i is tile i + 2: two tile times to land.
//   ONE image serves both roles of the latent row.  As keys it is read with ds_read_b128 (lane = row, A operand of
//     S^T = K Q^T), as values with ds_read_b64_tr_b16 (A operand of O^T += V^T P^T).  The image is lane-linear, so the
//     bank spread is made on the SOURCE side: lane (row, pos) fetches chunk pos ^ g(row) of its plane row,
//     g(row) = ((row & 3) << 2) | ((row >> 2) & 3) -- the 16 rows of a ds_read_b128 lane group land in 16 different
//     chunks, and the four rows of a transposing read in four different 64-byte groups (both conflict-free).
//   Per tile and wave: 36 + 32 MFMAs 32x32x16 (2176 cycles of the SIMD's matrix pipe), 16 exponentials per lane; LDS
//     fragment reads run six MFMAs ahead (inline asm, counted lgkmcnt), the DMA pieces, index loads and address updates
//     ride between the MFMAs.  Measured (profiles/r02_pmc_mla_shared.txt): zero LDS bank conflicts, LDS 17 % busy,
//     matrix pipe 29-33 % busy -- one wave per SIMD has nobody to hide its softmax, barrier and register shuffles behind.
//
// Arithmetic: base-2 online softmax with the scale folded into the exponent's fma, the running maximum moved only when
// it grows by more than 2^6 (extend_attention_shared_kv.hip); P in the activation type.  Rows in the activation type,
// no logit cap, heads a multiple of 32 * NWV; everything else stays on mla_decode_attention.hip.
// Replaces _fwd_grouped_kernel_stage1 for Lk = 576 / Lv = 512 (layers/attention/triton_ops/decode_attention.py:234-390).
#include "common.h"
#include "mfma_frag.h"

#include <cstdlib>
#include <type_traits>

namespace semipd {

namespace mls {

constexpr int kDK = 576, kDV = 512, kTok = 32, kRing = 3;
constexpr int kPlane = kTok * 256;                 // 8 KiB: 32 rows x 256 B
constexpr int kRope = 4 * kPlane;                  // the rope plane (32 rows x 128 B) behind the four planes
constexpr int kStage = 4 * kPlane + kTok * 128;    // 36 KiB
constexpr int kQReg = 24, kQLds = 36 - kQReg;      // Q^T fragments of a wave: 24 in registers, every third one (12) in LDS
constexpr bool q_in_lds(int ks) { return ks % 3 == 2; }
constexpr int q_reg_index(int ks) { return ks - (ks + 1) / 3; }
constexpr int kQBase = kRing * kStage;             // 108 KiB of ring, then 12 KiB of Q^T per wave
constexpr int lds_bytes(int nwv) { return kQBase + nwv * kQLds * 1024; }   // 156 KiB with four waves
// LDS reads in flight behind k-step KS of S^T when fragments run kAhead steps ahead: one per step for the keys and one
// more in the steps that take Q^T from LDS.  (A ds_read under four waves' load takes ~200 cycles, an MFMA 32: three
// steps ahead left the matrix pipe waiting in every step.)
constexpr int kAhead = 6;
constexpr int reads_behind(int ks) {
  int n = 0;
  for (int j = ks + 1; j <= ks + kAhead && j < 36; ++j) n += 1 + (q_in_lds(j) ? 1 : 0);
  return n;
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int OFF> __device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ s16x4 lds_read_tr8(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace mls

#ifdef MLS_TRACE
// debug build only (make CXXFLAGS+=-DMLS_TRACE): cycles per phase of the tile loop, summed by wave 0 of workgroup 0
__device__ unsigned long long mls_trace_buf[8];
#define MLS_T(i)                                                             \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();            \
    if (blockIdx.x == 0 && tid == 0) mls_trace_buf[i] += now_ - t_prev_;      \
    t_prev_ = now_;                                                          \
  } while (0)
#else
#define MLS_T(i) do {} while (0)
#endif

template <typename T, int NWV>
__global__ void __launch_bounds__(64 * NWV, 1)
mla_decode_shared_kernel(T* __restrict__ out, const T* __restrict__ q, const T* __restrict__ kv_buf,
                         const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices,
                         float* __restrict__ attn_logits, int num_q_heads, int head_groups, int64_t q_stride,
                         int64_t o_stride, int64_t kvbuf_stride, int num_kv_splits, float sm_scale) {
  using namespace mls;
  constexpr int RPW = kTok / NWV;          // rows of a tile this wave fetches: 8 / 16
  constexpr int PP = RPW / 4;              // 1-KiB pieces per 256-byte plane: 2 / 4
  constexpr int PR = RPW / 8;              // 1-KiB pieces of the rope plane: 1 / 2
  constexpr int kDma = 4 * PP + PR;        // DMA instructions per tile and wave: 9 / 18
  extern __shared__ __attribute__((aligned(16))) char mls_smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)mls_smem;

  const int split = (int)blockIdx.x % num_kv_splits;
  const int tmp = (int)blockIdx.x / num_kv_splits;
  const int hg = tmp % head_groups, b = tmp / head_groups;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, hi = lane >> 5;
  const int h = (hg * NWV + wave) * 32 + col;

  // (readfirstlane: opaque scalars -- see extend_attention_shared_kv.hip)
  const int kv_start = __builtin_amdgcn_readfirstlane(kv_indptr[b]);
  const int seq_len = __builtin_amdgcn_readfirstlane(kv_indptr[b + 1]) - kv_start;
  const int per_split = (seq_len + num_kv_splits - 1) / num_kv_splits;
  const int s_begin = per_split * split;
  const int s_end = min(s_begin + per_split, seq_len);
  if (s_end <= s_begin) {
    if (num_kv_splits == 1) {
      uint4* orow = reinterpret_cast<uint4*>(out + (int64_t)b * o_stride + (int64_t)h * kDV + hi * 256);
      for (int i = 0; i < 32; ++i) orow[i] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const int n_tiles = (s_end - s_begin + kTok - 1) / kTok;

  // ---- Q^T fragments (B operand of S^T): lane (head, hi) holds Q[head][ks*16 + hi*8 .. +8] ----
  //      fragments kQReg.. live in this wave's own LDS block, fragment-major and lane-linear (the register file holds O^T,
  //      256, and cannot hold all 144 registers of Q^T next to the rest without spilling)
  Frag16 qf[kQReg];
  const uint32_t qx_addr = lds0 + kQBase + wave * (kQLds * 1024) + lane * 16;
  {
    const T* qrow = q + (int64_t)b * q_stride + (int64_t)h * kDK + hi * 8;
    uint4 tail[kQLds];
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) {
      if (q_in_lds(ks)) tail[ks / 3] = *reinterpret_cast<const uint4*>(qrow + ks * 16);
      else qf[q_reg_index(ks)].u = *reinterpret_cast<const uint4*>(qrow + ks * 16);
    }
#pragma unroll
    for (int i = 0; i < kQLds; ++i)
      *reinterpret_cast<uint4*>(mls_smem + kQBase + wave * (kQLds * 1024) + i * 1024 + lane * 16) = tail[i];
  }

  // ---- DMA duty of this wave: rows wave*RPW + j*4 + (lane >> 4) of every plane, rows wave*RPW + j*8 + (lane >> 3) of
  //      the rope plane; the element offset inside the row carries the swizzle ----
  int swzp[4] = {}, swzr[2] = {};
#pragma unroll
  for (int j = 0; j < PP; ++j) {
    const int r = wave * RPW + j * 4 + (lane >> 4);
    swzp[j] = ((lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3))) * 8;
  }
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    const int r = wave * RPW + j * 8 + (lane >> 3);
    swzr[j] = kDV + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  const int32_t* idx_base = kv_indices + kv_start;
  auto load_idx = [=](int it, int32_t (&ip)[4], int32_t (&ir)[2]) __attribute__((always_inline)) {
    const int n0 = s_begin + it * kTok + wave * RPW;
#pragma unroll
    for (int j = 0; j < PP; ++j) ip[j] = idx_base[min(n0 + j * 4 + (lane >> 4), s_end - 1)];   // past the end: the last row (masked)
#pragma unroll
    for (int j = 0; j < PR; ++j) ir[j] = idx_base[min(n0 + j * 8 + (lane >> 3), s_end - 1)];
  };
  // the same loads for the tile loop, invisible to the compiler's wait counting: it cannot count across the guarded DMA
  // pieces that follow and falls back to vmcnt(0) in front of the first use, which makes every tile land within the
  // iteration that requested it.  The loop waits by hand (vmcnt(kDma): everything older than the DMA pieces).
  auto load_idx_async = [=](int it, int32_t (&ip)[4], int32_t (&ir)[2]) __attribute__((always_inline)) {
    const int n0 = s_begin + it * kTok + wave * RPW;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const int32_t* a = idx_base + min(n0 + j * 4 + (lane >> 4), s_end - 1);
      asm volatile("global_load_dword %0, %1, off" : "=v"(ip[j]) : "v"(a));
    }
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      const int32_t* a = idx_base + min(n0 + j * 8 + (lane >> 3), s_end - 1);
      asm volatile("global_load_dword %0, %1, off" : "=v"(ir[j]) : "v"(a));
    }
  };
  auto issue_tile = [=](int it, const int32_t (&ip)[4], const int32_t (&ir)[2]) __attribute__((always_inline)) {
    const uint32_t dst = lds0 + (uint32_t)(it % kRing) * kStage;
#pragma unroll
    for (int j = 0; j < PP; ++j) {
      const T* p = kv_buf + ((int64_t)ip[j] * kvbuf_stride + swzp[j]);
#pragma unroll
      for (int pl = 0; pl < 4; ++pl)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + pl * 128),
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(dst + pl * kPlane + (wave * RPW + j * 4) * 256),
                                         16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      const T* p = kv_buf + ((int64_t)ir[j] * kvbuf_stride + swzr[j]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(dst + kRope + (wave * RPW + j * 8) * 128),
                                       16, 0, 0);
    }
  };

  // prologue: tiles 0 and 1 leave together with the Q loads
  int32_t ip_cur[4] = {}, ir_cur[2] = {}, ip_nxt[4] = {}, ir_nxt[2] = {};
  {
    int32_t ip0[4] = {}, ir0[2] = {}, ip1[4] = {}, ir1[2] = {};
    load_idx(0, ip0, ir0);
    if (1 < n_tiles) load_idx(1, ip1, ir1);
    if (2 < n_tiles) load_idx(2, ip_cur, ir_cur);
    issue_tile(0, ip0, ir0);
    if (1 < n_tiles) issue_tile(1, ip1, ir1);
  }
  // the Q loads and the indices complete HERE (left pending, their wait lands in the loop as vmcnt(0) and drains the ring)
  wait_vm<0>();
  wait_lgkm<0>();   // Q^T tail written
#pragma unroll
  for (int ks = 0; ks < kQReg; ++ks) asm volatile("" : "+v"(qf[ks].w[0]), "+v"(qf[ks].w[1]), "+v"(qf[ks].w[2]), "+v"(qf[ks].w[3]));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ip_cur[j]));
#pragma unroll
  for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(ir_cur[j]));

  // ---- fragment addresses in stage 0 (moved from stage to stage at the end of every tile; plane and k-step offsets
  //      are immediates) ----
  // keys, A operand of S^T: lane (row = col, hi) reads chunk (ks*2 + hi) of the row: plane ks >> 3, position
  // ((ks & 7)*2 + hi) ^ g(row); rope chunks (ks - 32)*2 + hi at position ^ ((row >> 1) & 7) of the 128-byte row
  uint32_t ka[8], kr[4];
  {
    const int g = ((col & 3) << 2) | ((col >> 2) & 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) ka[i] = lds0 + col * 256 + (((i * 2 + hi) ^ g) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) kr[i] = lds0 + kRope + col * 128 + (((i * 2 + hi) ^ ((col >> 1) & 7)) * 16);
  }
  // values, A operand of O^T through the transposing read: 16-lane group supplies dv block g1*16 of the 32-wide tile
  // t of a plane; lane gi of the group reads row 8e + 4hi + (gi >> 2), columns 4*(gi & 3) .. +3 (extend_attention_shared_kv.hip)
  uint32_t va0[4], va1[4];
  {
    const int gi = lane & 15, g1 = (lane >> 4) & 1, vr = gi >> 2;
    const int low = g1 * 2 + ((gi & 3) >> 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      va0[t] = lds0 + (4 * hi + vr) * 256 + ((((t ^ vr) << 2) | (low ^ hi)) * 16) + (gi & 1) * 8;
      va1[t] = lds0 + (8 + 4 * hi + vr) * 256 + ((((t ^ vr) << 2) | (low ^ (2 + hi))) * 16) + (gi & 1) * 8;
    }
  }

  f32x16 o_acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // running maximum in the log2 domain
  const float qk_scale = sm_scale * 1.4426950408889634f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // One piece of the DMA of a tile: pieces 0 .. 4*PP-1 are (row group j, plane), the rest the rope plane.  Issued
  // one per k-step inside S^T = K Q^T so that the address arithmetic runs under the MFMAs.
  const T* dma_p = kv_buf;
  auto dma_piece = [&](auto i_c, uint32_t dst) __attribute__((always_inline)) {
    constexpr int I = decltype(i_c)::value;
    if constexpr (I < 4 * PP) {
      constexpr int J = I >> 2, PL = I & 3;
      if constexpr (PL == 0) dma_p = kv_buf + ((int64_t)ip_cur[J] * kvbuf_stride + swzp[J]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_p + PL * 128),
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(dst + PL * kPlane + (wave * RPW + J * 4) * 256),
                                       16, 0, 0);
    } else {
      constexpr int J = I - 4 * PP;
      const T* pr = kv_buf + ((int64_t)ir_cur[J] * kvbuf_stride + swzr[J]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pr,
                                       (__attribute__((address_space(3))) void*)(uintptr_t)(dst + kRope + (wave * RPW + J * 8) * 128),
                                       16, 0, 0);
    }
  };

#ifdef MLS_TRACE
  unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
  int stage = 0;      // it % kRing
  int dprev = 0;      // what moves the value addresses from the stage of tile it - 1 to that of tile it
  for (int it = 0; it < n_tiles; ++it) {
    // tile `it` has landed (every wave waits for its own pieces, the barrier makes them everybody's) and everyone is
    // done with tile it - 1, whose stage takes tile it + 2
    MLS_T(0);
    if (it + 1 < n_tiles) wait_vm<kDma>(); else wait_vm<0>();
    MLS_T(1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    MLS_T(2);
    const bool do_dma = it + 2 < n_tiles;
    const uint32_t dma_dst = lds0 + (uint32_t)(stage == 0 ? kRing - 1 : stage - 1) * kStage;   // stage of tile it + 2 = it - 1
    const int delta = (stage == kRing - 1) ? -(kRing - 1) * kStage : kStage;

    // ---- S^T = K Q^T: 36 k-steps, fragments kAhead steps ahead.  One accumulator: the MFMAs of this kernel write
    //      accumulation registers only (512-register budget), all 256 of them hold O^T, so S^T displaces one tile of
    //      O^T for the length of this phase -- a second accumulator displaces a second tile (32 more register moves).
    //      Riding along between the MFMAs: the index loads for tile it + 3, the DMA of tile it + 2, the stage change
    //      of the value addresses ----
    f32x16 s0;
    {
      Frag16 kf[8], qx[4];
      auto rd = [&](auto ks_c) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_c)::value;
        if constexpr (KS < 32) kf[KS & 7].u = lds_read16<(KS >> 3) * kPlane>(ka[KS & 7]);
        else kf[KS & 7].u = lds_read16<0>(kr[KS - 32]);
        if constexpr (q_in_lds(KS)) qx[(KS / 3) & 3].u = lds_read16<(KS / 3) * 1024>(qx_addr);
      };
      static_for<0, kAhead>([&](auto ks) { rd(ks); });
      static_for<0, 36>([&](auto ks) {
        constexpr int KS = decltype(ks)::value;
        if constexpr (KS + kAhead < 36) rd(std::integral_constant<int, KS + kAhead>{});
        wait_lgkm<reads_behind(KS)>();
        const auto bq = q_in_lds(KS) ? as_frag<T>(qx[(KS / 3) & 3]) : as_frag<T>(qf[q_in_lds(KS) ? 0 : q_reg_index(KS)]);
        s0 = Mfma<T>::mma(as_frag<T>(kf[KS & 7]), bq, KS == 0 ? zero16 : s0);
        if constexpr (KS == 0) {
          if (it + 3 < n_tiles) load_idx_async(it + 3, ip_nxt, ir_nxt);
        } else if constexpr (KS <= kDma) {
          if (do_dma) dma_piece(std::integral_constant<int, KS - 1>{}, dma_dst);
        } else if constexpr (KS >= 20 && KS < 24) {
          va0[KS - 20] += dprev;
          va1[KS - 20] += dprev;
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    MLS_T(3);
    // the first value fragments leave now: their latency passes under the softmax
    Frag16 vf[8];
    auto rdv = [&](auto g_c) __attribute__((always_inline)) {
      constexpr int G = decltype(g_c)::value, S = G >> 4, TT = G & 15;
      constexpr int OFF = (TT >> 2) * kPlane + S * 16 * 256;
      vf[G & 7].s[0] = lds_read_tr8<OFF>(va0[TT & 3]);
      vf[G & 7].s[1] = lds_read_tr8<OFF>(va1[TT & 3]);
    };
    static_for<0, kAhead>([&](auto g) { rdv(g); });

    // ---- softmax of the 32 x 32 tile: accumulator slot r of lane (head, hi) is row n0 + (r & 3) + 8*(r >> 2) + 4*hi ----
    f32x16 sc = s0;
    const int n0 = s_begin + it * kTok;
    if (n0 + kTok > s_end) {
      const int thr = s_end - 1 - n0 - 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = ((r & 3) + 8 * (r >> 2)) <= thr ? sc[r] : -INFINITY;
    }
    bool resc = false;
    float alpha = 1.f;
    {
      float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[8], sc[9]), fmaxf(sc[10], sc[11])), fmaxf(fmaxf(sc[12], sc[13]), fmaxf(sc[14], sc[15]))));
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * qk_scale;   // qk_scale > 0; -inf stays -inf
      constexpr float kDeferLog2 = 6.f;
      const bool grow = (m_run == -INFINITY) ? (mx > -INFINITY) : (mx > m_run + kDeferLog2);
      resc = __any(grow);
      if (resc) {
        const float m_new = fmaxf(m_run, mx);
        alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
        m_run = m_new;
        // O^T *= alpha on the accumulation registers themselves.  (A C++ multiply makes O^T a vector-register value
        // for the allocator, which then keeps Q^T in scratch for the whole loop; a test of `resc` in front of each
        // tile's first MFMA instead of this one block costs ~50 cycles per test, 770 per tile.)
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float acc = o_acc[t][r], tmp;
            asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 %0, %1"
                         : "+a"(acc), "=&v"(tmp)
                         : "v"(alpha));
            o_acc[t][r] = acc;
          }
      }
    }
    // P^T: rows 0-15 of the tile (the first k-step of O^T += V^T P^T) now, rows 16-31 between the MFMAs of that k-step
    Frag16 pfr[2];
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
    float psum = 0.f;
    auto exp_quad = [&](auto qd_c) __attribute__((always_inline)) {
      constexpr int QD = decltype(qd_c)::value;
      const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[QD * 4], qk_scale, -m_use));      // exp2(-inf) = 0
      const float p1 = __builtin_amdgcn_exp2f(fmaf(sc[QD * 4 + 1], qk_scale, -m_use));
      const float p2 = __builtin_amdgcn_exp2f(fmaf(sc[QD * 4 + 2], qk_scale, -m_use));
      const float p3 = __builtin_amdgcn_exp2f(fmaf(sc[QD * 4 + 3], qk_scale, -m_use));
      psum += (p0 + p1) + (p2 + p3);
      uint32_t w0 = pack2<T>(p0, p1), w1 = pack2<T>(p2, p3);
      asm volatile("" : "+v"(w0), "+v"(w1), "+v"(psum));   // pinned here (the optimiser sinks the exponentials to their use)
      pfr[QD >> 1].w[(QD & 1) * 2] = w0;
      pfr[QD >> 1].w[(QD & 1) * 2 + 1] = w1;
    };
    exp_quad(std::integral_constant<int, 0>{});
    exp_quad(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    MLS_T(4);

    // ---- O^T += V^T P^T: two k-steps x 16 dv tiles, fragments kAhead steps ahead; the key addresses change stage
    //      between the MFMAs of the second k-step ----
    static_for<0, 32>([&](auto g) {
      constexpr int G = decltype(g)::value, S = G >> 4, TT = G & 15;
      if constexpr (G + kAhead < 32) {
        rdv(std::integral_constant<int, G + kAhead>{});
        wait_lgkm<2 * kAhead>();
      } else {
        wait_lgkm<2 * (31 - G)>();
      }
      o_acc[TT] = Mfma<T>::mma(as_frag<T>(vf[G & 7]), as_frag<T>(pfr[S]), o_acc[TT]);
      if constexpr (G == 2) exp_quad(std::integral_constant<int, 2>{});
      if constexpr (G == 6) exp_quad(std::integral_constant<int, 3>{});
      if constexpr (G >= 16 && G < 24) ka[G - 16] += delta;
      if constexpr (G >= 24 && G < 28) kr[G - 24] += delta;
      __builtin_amdgcn_sched_barrier(0);
    });
    l_run += psum;
    MLS_T(5);

    // ---- to the next stage; the indices fetched in the first k-step are due now (their wait leaves tile it + 2 in flight) ----
    dprev = delta;
    stage = (stage == kRing - 1) ? 0 : stage + 1;
    if (do_dma) wait_vm<kDma>(); else wait_vm<0>();
    MLS_T(6);
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ip_nxt[j]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(ir_nxt[j]));
#pragma unroll
    for (int j = 0; j < 4; ++j) ip_cur[j] = ip_nxt[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) ir_cur[j] = ir_nxt[j];
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: O^T slot (t, r) of lane (head, hi) is dv t*32 + (r & 3) + 8*(r >> 2) + 4*hi ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (num_kv_splits == 1) {
    T* orow = out + (int64_t)b * o_stride + (int64_t)h * kDV + 4 * hi;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 w;
        w.x = pack2<T>(o_acc[t][r4 * 4] * inv, o_acc[t][r4 * 4 + 1] * inv);
        w.y = pack2<T>(o_acc[t][r4 * 4 + 2] * inv, o_acc[t][r4 * 4 + 3] * inv);
        *reinterpret_cast<uint2*>(orow + t * 32 + 8 * r4) = w;
      }
  } else {
    float* dst = attn_logits + (((int64_t)b * num_q_heads + h) * num_kv_splits + split) * (kDV + 1);
    // four consecutive floats per store (rows of 513 floats: dword-aligned only, which global stores accept)
    struct __attribute__((packed, aligned(4))) F4 { float v[4]; };
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        F4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w.v[j] = o_acc[t][r4 * 4 + j] * inv;
        *reinterpret_cast<F4*>(dst + t * 32 + 8 * r4 + 4 * hi) = w;
      }
    // the running maximum is kept in the log2 domain: natural-log lse for the stage-2 merge
    if (hi == 0) dst[kDV] = m_run * 0.6931471805599453f + __logf(l_tot);
  }
}

// Returns 0 after a launch, -1 when the shape stays on mla_decode_attention.hip.
template <typename T>
int launch_mla_decode_shared(T* out, const T* q, const T* kv_buf, const int32_t* kv_indptr, const int32_t* kv_indices,
                             float* attn_logits, int64_t batch, int Hq, int64_t q_stride, int64_t o_stride,
                             int64_t kvbuf_stride, int splits, float sm_scale, float logit_cap, hipStream_t st) {
  // SEMIPD_MLA_SHARED: 0 = never, 2 = whenever the shape allows (tests), otherwise by size; read per call
  const char* mode_env = std::getenv("SEMIPD_MLA_SHARED");
  const bool enabled = !(mode_env && mode_env[0] == '0'), force = mode_env && mode_env[0] == '2';
  if (!enabled || logit_cap > 0.f || Hq < 64 || Hq % 64 != 0) return -1;
  if (((uintptr_t)q | (uintptr_t)kv_buf | (uintptr_t)out) & 15) return -1;
  if ((q_stride | o_stride | kvbuf_stride) & 7) return -1;
  const int nwv = (Hq % 128 == 0) ? 4 : 2;
  const int groups = Hq / (32 * nwv);
  const int64_t total = batch * groups * splits;
  // a workgroup loads 147 KB of Q^T and sits through three dependent memory latencies before its first tile: below
  // ~5/8 of the CUs' worth of workgroups the wide kernel's 2-4x as many, lighter ones finish first (B = 32, ctx 1.1 k,
  // 4 splits: 52 vs 60 us; B = 1, ctx 8 k: 48 vs 55; profiles/r02_kbench_mla_decode_short_contexts.txt).  The host
  // side picks its split count with the same number (layers/attention_backend.py: MLA_SHARED_MIN_WORKGROUPS).
  if (total < 160 && !force) return -1;
  if (total > 0x7fffffff) {
    set_error("mla_decode: grid too large");
    return SEMIPD_EINVAL;
  }
  if (total == 0) return 0;
  static std::atomic<uint64_t> lds_ok4{0}, lds_ok2{0};   // one bit per device
  if (nwv == 4) {
    auto kern = mla_decode_shared_kernel<T, 4>;
    if (ensure_dynamic_lds((const void*)kern, mls::lds_bytes(4), lds_ok4, "mla_decode_shared")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), mls::lds_bytes(4), st, out, q, kv_buf, kv_indptr, kv_indices,
                       attn_logits, Hq, groups, q_stride, o_stride, kvbuf_stride, splits, sm_scale);
  } else {
    auto kern = mla_decode_shared_kernel<T, 2>;
    if (ensure_dynamic_lds((const void*)kern, mls::lds_bytes(2), lds_ok2, "mla_decode_shared")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(128), mls::lds_bytes(2), st, out, q, kv_buf, kv_indptr, kv_indices,
                       attn_logits, Hq, groups, q_stride, o_stride, kvbuf_stride, splits, sm_scale);
  }
  return launch_status("mla_decode_shared");
}

#ifdef MLS_TRACE
extern "C" int semipd_debug_mls_trace(unsigned long long* out, int reset) {
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mls_trace_buf), sizeof(unsigned long long) * 8);
  if (reset) {
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mls_trace_buf), z, sizeof(z));
  }
  return 0;
}
#endif

template int launch_mla_decode_shared<bf16_t>(bf16_t*, const bf16_t*, const bf16_t*, const int32_t*, const int32_t*, float*,
                                              int64_t, int, int64_t, int64_t, int64_t, int, float, float, hipStream_t);
template int launch_mla_decode_shared<f16_t>(f16_t*, const f16_t*, const f16_t*, const int32_t*, const int32_t*, float*,
                                             int64_t, int, int64_t, int64_t, int64_t, int, float, float, hipStream_t);

}  // namespace semipd
