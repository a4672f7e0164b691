// Prefill ("extend") attention, shared-KV form for gfx950 (round 2).
//
// The first kernel (extend_attention.hip) gives every (q head, 128 tokens) its own workgroup: the four q heads of a
// GQA group each stage the same K / V rows through their own LDS tile (4x the fill and the L2 reads), every tile
// costs two __syncthreads around a register -> LDS copy, and one 1024-token request is 256 workgroups whose longest
// member walks 16 KV tiles alone.  This one is built the other way round:
//
//   workgroup = 8 waves = G q heads of ONE kv head x TB blocks of 32 tokens x 2 KV halves   (G * TB = 4)
//     wave (half, sub): q head = sub % G, token block = sub / G; per 128-row KV tile the waves of half 0 take rows
//     0-63 and those of half 1 rows 64-127, and the two partial (m, l, O) are merged through LDS at the end --
//     the walk of one row block is half as long, and a 1024-token Llama-3 request is 256 workgroups x 8 waves.
//   K / V tiles (128 rows x 256 B each) arrive by LDS-DMA (buffer_load ... lds for the new tokens, global_load ... lds
//     through kv_indices for the paged prefix) into two-deep rings: no staging registers, no ds_write.
//   Each tile is two blocks per wave -- a MATRIX block (O^T += V^T P^T of the previous tile, then S^T = K Q^T of this
//     one: 32 MFMAs) and a VECTOR block (the softmax) -- separated by s_barrier, and the waves of half 1 run one block
//     behind those of half 0: the two waves that share a SIMD are always in opposite blocks (ping-pong).  The DMA image is lane-linear (row-major, 16-B chunks); bank conflicts are removed on the SOURCE
//     side: lane (row, pos) fetches chunk pos ^ (row & 15) of a K row (ds_read_b128 fragments: 16 rows of a lane
//     group land in 16 different chunks) and chunk pos ^ ((row & 3) << 2) of a V row (ds_read_b64_tr_b16: the four
//     rows of a lane group land in four different 64-B groups).
//   Rows past the end of a tile are fetched from the last valid row instead (finite data, masked to p = 0).
//   Fragment reads are inline asm with counted lgkmcnt waits: a compiler-visible ds_read would wait vmcnt(0) for the
//     DMA of the NEXT tile that is in flight on purpose (see stream_linear.hip).
//
// Arithmetic is the first kernel's FAST path instruction for instruction (S^T = K Q^T, base-2 online softmax with
// the scale folded into the exponent's fma, O^T += V^T P^T with P in registers), so results agree with it to the
// rounding of the half-merge.  Exact head size 128, rows in the activation type, no logit cap; everything else stays
// on extend_attention.hip.   Mirrors extend_attention_fwd (layers/attention/triton_ops/extend_attention.py:291-410).
#include "common.h"
#include "mfma_frag.h"

#include <cstdlib>
#include <type_traits>

namespace semipd {

namespace skv {

constexpr int kD = 128;               // head size (q, k and v)
constexpr int kRowBytes = kD * 2;     // one K or V row in LDS
// KV rows per tile = 64 * HALVES; one K (or V) tile = 16 KiB * HALVES; K stages at 0 / stage, V stages at 2 / 3 x stage
constexpr int tile_rows(int halves) { return 64 * halves; }
constexpr int stage_bytes(int halves) { return tile_rows(halves) * kRowBytes; }
constexpr int lds_bytes(int halves) { return 4 * stage_bytes(halves); }   // 64 KiB (two workgroups per CU) / 128 KiB
constexpr int kMergeStride = 66 * 64 * 4;   // one wave's (O 64, m, l) x 64 lanes, fp32

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
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// (WAIT_VM0_AND) The prefetched pool indices are "used" right behind the explicit wait: the compiler then places its own wait for those
// loads here (where the counter is zero anyway) instead of in the middle of the next DMA issue, where a vmcnt(0) would
// sit on the pieces just sent.
}  // namespace skv

#define SKV_WAIT_VM0_AND(IDX)                                          \
  do {                                                                 \
    skv::wait_vm0();                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < kPPW; ++j_) asm volatile("" : "+v"((IDX)[j_])); \
  } while (0)

template <typename T, int G, int TB, int HALVES>
__global__ void __launch_bounds__(64 * G * TB * HALVES, (G * TB * HALVES == 8) ? 1 : (HALVES == 2 ? 1 : 2))
extend_attn_shared_kv_kernel(T* __restrict__ out, const T* __restrict__ q_ext, const T* __restrict__ k_ext,
                             const T* __restrict__ v_ext, const T* __restrict__ k_buf, const T* __restrict__ v_buf,
                             const int32_t* __restrict__ qo_indptr, const int32_t* __restrict__ kv_indptr,
                             const int32_t* __restrict__ kv_indices, int group, int64_t q_stride, int64_t k_stride,
                             int64_t v_stride, int64_t o_stride, int64_t kbuf_stride, int64_t vbuf_stride,
                             float sm_scale) {
  using namespace skv;
  // G q heads x TB blocks of 32 tokens x HALVES wave groups = NW waves; every wave sends kPPW 1-KiB pieces of a K or V tile
  constexpr int NW = G * TB * HALVES, kGT = G * TB, kPPW = 16 / kGT, kWaveRows = 4 * kPPW;
  static_assert(kGT == 4 || kGT == 2 || kGT == 1, "4, 2 or 1 (head, token block) pairs per wave group");
  constexpr int kTileRows = tile_rows(HALVES), kStage = stage_bytes(HALVES), kVBase = 2 * kStage;
  extern __shared__ __attribute__((aligned(16))) char skv_smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)skv_smem;

  const int seq = blockIdx.z;
  const int hpg = group / G;           // workgroups per kv head
  const int hk = (int)blockIdx.x / hpg, hgrp = (int)blockIdx.x - hk * hpg;
  const int qt = (int)(gridDim.y - 1 - blockIdx.y);     // longest (last) token tile first
  // readfirstlane: opaque SGPR values -- otherwise the compiler re-loads them from memory inside the tile loop (it
  // assumes the LDS-DMA builtins may have written anywhere), and that load's wait drains the DMA in flight
  const int q_start = __builtin_amdgcn_readfirstlane(qo_indptr[seq]);
  const int ext_len = __builtin_amdgcn_readfirstlane(qo_indptr[seq + 1]) - q_start;
  if (qt * TB * 32 >= ext_len) return;
  const int kv_start = __builtin_amdgcn_readfirstlane(kv_indptr[seq]);
  const int pre_len = __builtin_amdgcn_readfirstlane(kv_indptr[seq + 1]) - kv_start;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = HALVES == 2 ? (wave / kGT) : 0, sub = wave % kGT;
  const int hq = hk * group + hgrp * G + sub % G;
  const int q0 = (qt * TB + sub / G) * 32;               // this wave's first query row
  const int col = lane & 31, hi = lane >> 5;
  const int q_local = q0 + col;
  const bool q_valid = q_local < ext_len;
  const bool wave_active = q0 < ext_len;

  // ---- Q^T fragments (B operand): lane holds Q[q_local][ks*16 + hi*8 .. +8] ----
  Frag16 qf[8];
  {
    const T* qrow = q_ext + (int64_t)(q_start + q_local) * q_stride + (int64_t)hq * kD;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      qf[ks].u = q_valid ? *reinterpret_cast<const uint4*>(qrow + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // running max in the log2 domain
  const float qk_scale = sm_scale * 1.4426950408889634f;

  // ---- the tile walk: paged prefix (no causal mask), then the new tokens up to this workgroup's last row ----
  const int ext_end = min(ext_len, (qt + 1) * TB * 32);
  const int n_pre = (pre_len + kTileRows - 1) / kTileRows;
  const int n_ext = (ext_end + kTileRows - 1) / kTileRows;
  const int n_tiles = n_pre + n_ext;

  // DMA duty of this wave: rows wave*16 + j*4 + (lane >> 4), j = 0..3, of a K tile or of a V tile; pos = lane & 15
  const int drow = wave * kWaveRows + (lane >> 4), dpos = lane & 15;
  const T* k_pre = k_buf + (int64_t)hk * kD;
  const T* v_pre = v_buf + (int64_t)hk * kD;
  const T* ke_head = k_ext + (int64_t)q_start * k_stride + (int64_t)hk * kD;
  const T* ve_head = v_ext + (int64_t)q_start * v_stride + (int64_t)hk * kD;
  const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
      (void*)ke_head, 0, (int)(((int64_t)(ext_end - 1) * k_stride + kD) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
      (void*)ve_head, 0, (int)(((int64_t)(ext_end - 1) * v_stride + kD) * 2), 0x00020000);
  const int32_t* idx_base = kv_indices + kv_start;
  // pool slots of this lane's rows: idx_k for the K tile about to be fetched, idx_v (the previous idx_k) for the V tile
  // that follows one tile behind; loaded one step ahead so that the index -> address chain is never waited for
  // (arrays of the fixed maximum size: an array whose size depends on the template arguments, captured by the lambdas
  // below, makes the HOST pass drop the kernel's stub without a diagnostic; unused elements cost nothing)
  int32_t idx_k[16] = {}, idx_v[16] = {};
  auto load_idx = [=, &idx_k](int it) __attribute__((always_inline)) {
    if (it < n_pre) {
#pragma unroll
      for (int j = 0; j < kPPW; ++j) idx_k[j] = idx_base[min(it * kTileRows + drow + j * 4, pre_len - 1)];
    }
  };
  // whole tiles of new tokens: per-lane byte offsets computed once, the tile offset is the scalar operand of the buffer load
  int kvo[16], vvo[16];
#pragma unroll
  for (int j = 0; j < kPPW; ++j) {
    const int r = drow + j * 4;
    kvo[j] = r * (int)k_stride * 2 + (dpos ^ (r & 15)) * 16;          // the launcher checked ext rows * stride < 2^31 bytes
    vvo[j] = r * (int)v_stride * 2 + (dpos ^ ((r & 3) << 2)) * 16;
  }
  // one tile (K: IS_V = false, swizzle row & 15; V: swizzle (row & 3) << 2) into stage (it & 1)
  auto issue_tile = [=](auto is_v, int it, const int32_t (&idx)[16]) __attribute__((always_inline)) {
    constexpr bool IS_V = decltype(is_v)::value;
    const uint32_t dst = lds0 + (IS_V ? kVBase : 0) + (it & 1) * kStage + wave * kWaveRows * kRowBytes;
    const int n0 = (it - n_pre) * kTileRows;
    if (it >= n_pre && n0 + kTileRows <= ext_end) {
      // (readfirstlane: the compiler cannot prove the tile offset uniform and would wrap every load in a waterfall loop)
      const int tile_off = __builtin_amdgcn_readfirstlane(n0 * (int)(IS_V ? v_stride : k_stride) * 2);
#pragma unroll
      for (int j = 0; j < kPPW; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_V ? rsrc_v : rsrc_k,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(dst + j * 4 * kRowBytes),
                                                 16, IS_V ? vvo[j] : kvo[j], tile_off, 0, 0);
      return;
    }
    // the paged prefix and the last, partial tile of new tokens: addresses from scratch each time (an opaque copy of the
    // lane id keeps the compiler from hoisting a dozen 64-bit per-lane constants out of the loop and spilling them)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int rw = wave * kWaveRows + (ln >> 4), ps = ln & 15;
    if (it < n_pre) {
      const T* base = IS_V ? v_pre : k_pre;
      const int64_t stride = IS_V ? vbuf_stride : kbuf_stride;
#pragma unroll
      for (int j = 0; j < kPPW; ++j) {
        const int r = rw + j * 4;
        const int sw = IS_V ? ((r & 3) << 2) : (r & 15);
        const T* p = base + ((int64_t)idx[j] * stride + ((ps ^ sw) * 8));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(dst + j * 4 * kRowBytes), 16, 0, 0);
      }
    } else {
      const int stride_b = (int)(IS_V ? v_stride : k_stride) * 2;
#pragma unroll
      for (int j = 0; j < kPPW; ++j) {
        const int r = rw + j * 4;
        const int sw = IS_V ? ((r & 3) << 2) : (r & 15);
        const int n = min(n0 + r, ext_end - 1);      // rows past the end: the last valid row (finite, masked)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_V ? rsrc_v : rsrc_k,
                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(dst + j * 4 * kRowBytes),
                                                 16, n * stride_b + (ps ^ sw) * 16, 0, 0, 0);
      }
    }
  };

  // prologue, first part: the DMA of K(0) goes out while the Q rows are still in flight (one memory latency instead of two
  // in front of the first tile -- a single 1024-token request is 8 tile steps long, the prologue is a third of its time)
  // K(0), V(0) and K(1) all leave now (both stages of both rings are free): the first tile step then finds its V tile
  // landed instead of waiting one more memory latency for it
  load_idx(0);
  issue_tile(std::false_type{}, 0, idx_k);
  issue_tile(std::true_type{}, 0, idx_k);
  load_idx(1);
  if (1 < n_tiles) issue_tile(std::false_type{}, 1, idx_k);
#pragma unroll
  for (int j = 0; j < kPPW; ++j) idx_v[j] = idx_k[j];
  load_idx(2);
  // the Q loads complete HERE (before the lambdas below capture them): left pending into the loop, their wait lands in
  // front of the first MFMA of every tile as vmcnt(0) and drains the DMA of the next tile
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    asm volatile("" : "+v"(qf[ks].w[0]), "+v"(qf[ks].w[1]), "+v"(qf[ks].w[2]), "+v"(qf[ks].w[3]));

  // ---- fragment addresses (stage / kv-block offsets are immediates) ----
  // K, A operand of S^T: lane (col, hi) reads row half*64 + kt*32 + col, chunk (ks*2 + hi) ^ (col & 15)
  uint32_t kaddr[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    kaddr[ks] = lds0 + (half * 64 + col) * kRowBytes + (((ks * 2 + hi) ^ (col & 15)) * 16);
  // V^T, A operand of O^T, through the transposing read: 16-lane group g = lane >> 4 supplies dv block (g & 1) * 16 of
  // a 32-wide dv tile t; lane i of the group reads row 4*hi + (i >> 2), columns 4*(i & 3) .. +3.  The kv-slot
  // permutation of the S^T accumulator layout is matched by the row offsets of the two reads (rows +0 and +8).
  const int gi = lane & 15, g1 = (lane >> 4) & 1, vr = gi >> 2;
  uint32_t vaddr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    vaddr[t] = lds0 + kVBase + (half * 64 + hi * 4 + vr) * kRowBytes + ((t ^ vr) * 64 + g1 * 32 + (gi & 3) * 8);

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- the tile loop, software-pipelined inside the wave ----
  // In program order QK^T -> softmax -> PV of one tile is a dependent chain: the wave's 32 MFMAs (1024 cycles of its
  // SIMD's matrix pipe) and its ~200 VALU instructions (~1000 issue cycles) add up, and with every wave of the
  // workgroup in the same stage behind the tile barrier the partner wave on the SIMD does not fill the gaps either
  // (PMC of the first version: 4900 cycles per tile and SIMD; profiles/r02_pmc_extend_shared_kv.txt).  Here the chain
  // is cut across tiles: while the matrix pipe computes S^T of tile i + 1, the same wave turns S^T of tile i into P^T
  // (exp2 / sum / pack, in slices between the MFMAs), and while it accumulates O^T += V^T P^T of tile i the row maxima
  // of tile i + 1 are reduced.  K runs one tile ahead of V in the DMA rings.
  //   A(i): mask, row maxima of S^T(i), decide whether the running maximum moves (and then rescale O, l)
  //   B(i): P^T(i) = exp2(S^T(i) * scale - m), l += row sums; the packed words land in the first 16 registers of S^T(i)
  f32x16 s_a[2], s_b[2];
  float mx_part = -INFINITY;

  // masking of one S^T tile (boundary tiles only; wave-uniform test): kv row of accumulator slot (kt, r) =
  // n0 + kt*32 + (r & 3) + 8*(r >> 2) + 4*hi, visible when < n_end and, among the new tokens, <= this lane's query row
  auto mask_tile = [=](f32x16 (&sc)[2], bool pre, int n0, int n_end) __attribute__((always_inline)) {
    const bool need_mask = (n0 + 64 > n_end) || (!pre && n0 + 63 > q0);
    if (need_mask) {
      const int thr = (pre ? n_end - 1 : min(q_local, n_end - 1)) - 4 * hi;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sc[kt][r] = (n0 + kt * 32 + (r & 3) + 8 * (r >> 2)) <= thr ? sc[kt][r] : -INFINITY;
    }
  };
  // the end of A: combine the two 16-slot halves of a row (lane ^ 32), move the running maximum if it has to.
  // The maximum only moves when some row's grew by more than 2^kDeferLog2 (or was -inf): until then p = 2^(s - m) may
  // exceed 1 by at most that factor -- p is a floating-point type, its relative precision is unchanged, l and O carry
  // the same factor and it cancels in O / l.  Saves the 64-register rescale of O^T on almost every tile (random data:
  // some row of 32 finds a new maximum in 40 % of the tiles even 64 tiles in).  O, l and m move together, after the
  // previous tile's P^T V is complete and before this tile's P is formed.
  auto finish_max = [=, &o_acc, &m_run, &l_run](float mx, bool live) __attribute__((always_inline)) {
    constexpr float kDeferLog2 = 6.f;
    {   // v_permlane32_swap, no LDS round trip
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    mx *= qk_scale;   // qk_scale > 0; -inf stays -inf
    const bool grow = live && ((m_run == -INFINITY) ? (mx > -INFINITY) : (mx > m_run + kDeferLog2));
    if (__any(grow)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[t][r] *= alpha;
      m_run = m_new;
    }
  };
  // one slice of B: four scores of sc -> two packed words of P^T, in place
  auto exp_slice = [=](auto ks_c, f32x16 (&sc)[2], float m_use, float& psum) __attribute__((always_inline)) {
    constexpr int KS = decltype(ks_c)::value, KT = KS >> 2, R0 = (KS & 3) * 4;
    const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[KT][R0], qk_scale, -m_use));       // exp2(-inf) = 0
    const float p1 = __builtin_amdgcn_exp2f(fmaf(sc[KT][R0 + 1], qk_scale, -m_use));
    const float p2 = __builtin_amdgcn_exp2f(fmaf(sc[KT][R0 + 2], qk_scale, -m_use));
    const float p3 = __builtin_amdgcn_exp2f(fmaf(sc[KT][R0 + 3], qk_scale, -m_use));
    psum += (p0 + p1) + (p2 + p3);
    uint32_t w0 = pack2<T>(p0, p1), w1 = pack2<T>(p2, p3);
    // pinned HERE: the optimiser otherwise sinks all 32 exponentials to their use, behind the 16 MFMAs they are to hide in
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(psum));
    sc[0][KT * 8 + (R0 >> 1)] = __uint_as_float(w0);
    sc[0][KT * 8 + (R0 >> 1) + 1] = __uint_as_float(w1);
  };
  // S^T(next) = K Q^T from K stage KST, with B(cur) in slices between the MFMAs (WITH_B) or alone (the first tile)
  auto qk_phase = [=, &l_run](auto kst_c, auto with_b, f32x16 (&s_nxt)[2], f32x16 (&s_cur)[2], float m_use) __attribute__((always_inline)) {
    constexpr int KO = decltype(kst_c)::value * kStage;
    constexpr bool WITH_B = decltype(with_b)::value;
    Frag16 kf[2][2];   // [buffer][kv block]: the next k-step's fragments are read while this one multiplies
    float psum = 0.f;
    kf[0][0].u = lds_read16<KO>(kaddr[0]);
    kf[0][1].u = lds_read16<KO + 32 * kRowBytes>(kaddr[0]);
    static_for<0, 8>([&](auto ks) {
      constexpr int KS = decltype(ks)::value, CUR = KS & 1, NXT = CUR ^ 1;
      if constexpr (KS < 7) {
        kf[NXT][0].u = lds_read16<KO>(kaddr[KS + 1]);
        kf[NXT][1].u = lds_read16<KO + 32 * kRowBytes>(kaddr[KS + 1]);
        wait_lgkm<2>();
      } else {
        wait_lgkm<0>();
      }
      s_nxt[0] = Mfma<T>::mma(as_frag<T>(kf[CUR][0]), as_frag<T>(qf[KS]), KS == 0 ? zero16 : s_nxt[0]);
      s_nxt[1] = Mfma<T>::mma(as_frag<T>(kf[CUR][1]), as_frag<T>(qf[KS]), KS == 0 ? zero16 : s_nxt[1]);
      if constexpr (WITH_B) exp_slice(ks, s_cur, m_use, psum);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (WITH_B) l_run += psum;
  };
  // O^T += V^T P^T from V stage VST (P^T = the first 16 registers of s_cur), the row maxima of s_nxt in slices between
  // the MFMAs; returns this lane's partial maximum
  auto pv_phase = [=, &o_acc](auto vst_c, const f32x16 (&s_cur)[2], const f32x16 (&s_nxt)[2]) __attribute__((always_inline)) -> float {
    constexpr int VO = decltype(vst_c)::value * kStage;
    Frag16 pfr[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) pfr[i >> 1][i & 1].w[w] = __float_as_uint(s_cur[0][i * 4 + w]);
    // 8 steps = (kv block, slot) x (dv tiles 0-1 | 2-3): two MFMAs each, the fragments of the next step in flight
    Frag16 vf[2][2];
    float mxa = -INFINITY, mxb = -INFINITY;
    vf[0][0].s[0] = lds_read_tr8<VO>(vaddr[0]);
    vf[0][0].s[1] = lds_read_tr8<VO + 8 * kRowBytes>(vaddr[0]);
    vf[0][1].s[0] = lds_read_tr8<VO>(vaddr[1]);
    vf[0][1].s[1] = lds_read_tr8<VO + 8 * kRowBytes>(vaddr[1]);
    static_for<0, 8>([&](auto g) {
      constexpr int GG = decltype(g)::value;
      constexpr int GRP = GG >> 1, T0 = (GG & 1) * 2, CUR = GG & 1, NXT = CUR ^ 1;
      constexpr int KT = GRP >> 1, S2 = GRP & 1;
      if constexpr (GG < 7) {
        constexpr int NG = (GG + 1) >> 1, NT0 = ((GG + 1) & 1) * 2;
        constexpr int NO = VO + ((NG >> 1) * 32 + (NG & 1) * 16) * kRowBytes;
        vf[NXT][0].s[0] = lds_read_tr8<NO>(vaddr[NT0]);
        vf[NXT][0].s[1] = lds_read_tr8<NO + 8 * kRowBytes>(vaddr[NT0]);
        vf[NXT][1].s[0] = lds_read_tr8<NO>(vaddr[NT0 + 1]);
        vf[NXT][1].s[1] = lds_read_tr8<NO + 8 * kRowBytes>(vaddr[NT0 + 1]);
        wait_lgkm<4>();
      } else {
        wait_lgkm<0>();
      }
      o_acc[T0] = Mfma<T>::mma(as_frag<T>(vf[CUR][0]), as_frag<T>(pfr[KT][S2]), o_acc[T0]);
      o_acc[T0 + 1] = Mfma<T>::mma(as_frag<T>(vf[CUR][1]), as_frag<T>(pfr[KT][S2]), o_acc[T0 + 1]);
      constexpr int MK = GG >> 2, MR = (GG & 3) * 4;
      mxa = fmaxf(mxa, fmaxf(s_nxt[MK][MR], s_nxt[MK][MR + 1]));
      mxb = fmaxf(mxb, fmaxf(s_nxt[MK][MR + 2], s_nxt[MK][MR + 3]));
      asm volatile("" : "+v"(mxa), "+v"(mxb));
      __builtin_amdgcn_sched_barrier(0);
    });
    return fmaxf(mxa, mxb);
  };
  // tile geometry of this wave's half (scalars)
  auto tile_n0 = [=](int it) __attribute__((always_inline)) { return (it < n_pre ? it : it - n_pre) * kTileRows + half * 64; };

  // one iteration: [K(it+1), V(it) landed] barrier [DMA of K(it+2), V(it+1)] QK(it+1) || B(it), mask, PV(it) || max(it+1), A(it+1)
  auto iteration = [=, &o_acc, &m_run, &l_run, &idx_k, &idx_v](auto par_c, int it, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;   // it & 1
    SKV_WAIT_VM0_AND(idx_k);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (it + 2 < n_tiles) issue_tile(std::false_type{}, it + 2, idx_k);   // into K stage (it & 1): K(it) was read last iteration
    if (it + 1 < n_tiles) issue_tile(std::true_type{}, it + 1, idx_v);    // into V stage ((it + 1) & 1): V(it - 1) likewise
#pragma unroll
    for (int j = 0; j < kPPW; ++j) idx_v[j] = idx_k[j];
    load_idx(it + 3);
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;   // everything masked so far
    qk_phase(std::integral_constant<int, PAR ^ 1>{}, std::true_type{}, s_nxt, s_cur, m_use);
    const bool live = it + 1 < n_tiles;   // past the last tile S^T(it + 1) is computed from stale LDS and dropped
    if (live) {
      const bool pre = it + 1 < n_pre;
      mask_tile(s_nxt, pre, tile_n0(it + 1), pre ? pre_len : ext_len);
    }
    const float mx = pv_phase(std::integral_constant<int, PAR>{}, s_cur, s_nxt);
    finish_max(mx, live);
  };

  // prologue, second part: S^T(0) and A(0) without anything to overlap with
  SKV_WAIT_VM0_AND(idx_k);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  qk_phase(std::integral_constant<int, 0>{}, std::false_type{}, s_a, s_b, 0.f);
  {
    const bool pre = 0 < n_pre;
    mask_tile(s_a, pre, tile_n0(0), pre ? pre_len : ext_len);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_a[kt][r]);
    finish_max(mx, true);
  }
  (void)mx_part;
  for (int it = 0; it < n_tiles; it += 2) {
    iteration(std::integral_constant<int, 0>{}, it, s_a, s_b);
    if (it + 1 >= n_tiles) break;
    iteration(std::integral_constant<int, 1>{}, it + 1, s_b, s_a);
  }

  // ---- merge the two KV halves: the waves of half 1 hand (O, m, l) to their partners through LDS ----
  float* mbuf = reinterpret_cast<float*>(skv_smem + sub * kMergeStride);
  float m1 = -INFINITY, l1 = 0.f;
  if constexpr (HALVES == 2) {
  __syncthreads();   // every fragment read of the last tile is done
  if (half == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mbuf[(t * 16 + r) * 64 + lane] = o_acc[t][r];
    mbuf[64 * 64 + lane] = m_run;
    mbuf[65 * 64 + lane] = l_run;
  }
  __syncthreads();
  if (half == 1 || !wave_active) return;
  m1 = mbuf[64 * 64 + lane];
  l1 = mbuf[65 * 64 + lane];
  } else {
    if (!wave_active) return;
  }
  const float m_all = fmaxf(m_run, m1);
  const float a0 = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_all);
  const float a1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m1 - m_all);
  const float l_lane = l_run * a0 + l1 * a1;
  const float l_tot = l_lane + __shfl_xor(l_lane, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  const float s0 = a0 * inv, s1 = a1 * inv;
  if (q_valid) {
    T* orow = out + (int64_t)(q_start + q_local) * o_stride + (int64_t)hq * kD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = HALVES == 2 ? o_acc[t][r4 * 4 + j] * s0 + mbuf[(t * 16 + r4 * 4 + j) * 64 + lane] * s1 : o_acc[t][r4 * 4 + j] * s0;
        uint2 w;
        w.x = pack2<T>(o[0], o[1]);
        w.y = pack2<T>(o[2], o[3]);
        *reinterpret_cast<uint2*>(orow + t * 32 + 8 * r4 + 4 * hi) = w;
      }
    }
  }
}

// Returns 0 when launched, 1 when the shape is not covered (the caller falls back to extend_attention.hip).
// Forms (G heads x TB token blocks x HALVES wave groups), chosen by how many workgroups the batch gives:
//   plenty (> 512 of the 8-wave form): HALVES = 1 -- 4 waves, 64-row tiles, two workgroups per CU (the second one covers
//     the other's prologue and epilogue);
//   few: HALVES = 2 -- 8 waves, 128-row tiles split between two wave groups: the walk of a row block is half as long.
// (Measured and dropped: half the heads per workgroup for a single short request -- 4 waves, ONE per SIMD, twice the
// workgroups.  A lone wave per SIMD reaches about half the rate of two: 32.7 us against 26.9 for one 1024-token request.)
template <typename T>
int launch_extend_shared_kv(void* out, const void* q, const void* k, const void* v, const void* k_buf, const void* v_buf,
                            const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices, int64_t batch,
                            int Hq, int Hkv, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride,
                            int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend, float sm_scale, hipStream_t st) {
  const int group = Hq / Hkv;
  const int g8 = group % 4 == 0 ? 4 : (group % 2 == 0 ? 2 : 1);   // heads per workgroup of the 8-wave form
  const int64_t wgs8 = (int64_t)Hkv * (group / g8) * ((max_len_extend + (4 / g8) * 32 - 1) / ((4 / g8) * 32)) * batch;
  static const int force = [] { const char* e = getenv("SEMIPD_EXTEND_KV_FORM"); return e ? atoi(e) : 0; }();   // 1 / 2
  // the 8-wave form keeps ONE workgroup per CU: it pays while its workgroups are whole rounds (one or two) of the CUs this
  // process owns -- one 1024-token Llama-3 request on the whole chip (256), two (512); on a 160-CU share the same 256
  // workgroups are 1.6 rounds and the 4-wave form (two per CU, all resident) is faster at every size
  // (profiles/r03_kbench_extend_forms_160cu.txt: 1 x 1024 33.0 -> 31.4 us, 2 x 1024 behind prefixes 116 -> 100)
  const int own = owned_cus().load(std::memory_order_relaxed);
  const int64_t cus = own > 0 ? own : 256;
  const bool whole_rounds = wgs8 <= cus || (wgs8 <= 2 * cus && wgs8 % cus == 0);
  const int form = force == 1 || force == 2 ? force : (whole_rounds ? 2 : 1);
  const int G = g8, TB = 4 / g8;
  const unsigned gx = (unsigned)(Hkv * (group / G));
  const unsigned gy = (unsigned)((max_len_extend + TB * 32 - 1) / (TB * 32));
  if (gy > 65535u || gx > 65535u) return 1;
  dim3 grid(gx, gy, (unsigned)batch);
#define SKV(GV, TV, HV)                                                                                            \
  do {                                                                                                             \
    static std::atomic<uint64_t> lds_ok{0};                                                                         \
    if (ensure_dynamic_lds((const void*)extend_attn_shared_kv_kernel<T, GV, TV, HV>, skv::lds_bytes(HV), lds_ok,      \
                           "extend_attn_shared_kv"))                                                                  \
      return 1;                                                                                                       \
    hipLaunchKernelGGL((extend_attn_shared_kv_kernel<T, GV, TV, HV>), grid, dim3(64 * GV * TV * HV),                \
                       skv::lds_bytes(HV), st, (T*)out, (const T*)q, (const T*)k, (const T*)v, (const T*)k_buf,     \
                       (const T*)v_buf, qo_indptr, kv_indptr, kv_indices, group, q_stride, k_stride, v_stride,      \
                       o_stride, kbuf_stride, vbuf_stride, sm_scale);                                               \
  } while (0)
  if (form == 1) {
    if (g8 == 4) SKV(4, 1, 1);
    else if (g8 == 2) SKV(2, 2, 1);
    else SKV(1, 4, 1);
  } else {
    if (g8 == 4) SKV(4, 1, 2);
    else if (g8 == 2) SKV(2, 2, 2);
    else SKV(1, 4, 2);
  }
#undef SKV
  return 0;
}

template int launch_extend_shared_kv<bf16_t>(void*, const void*, const void*, const void*, const void*, const void*,
                                             const int32_t*, const int32_t*, const int32_t*, int64_t, int, int, int64_t,
                                             int64_t, int64_t, int64_t, int64_t, int64_t, int, float, hipStream_t);
template int launch_extend_shared_kv<f16_t>(void*, const void*, const void*, const void*, const void*, const void*,
                                            const int32_t*, const int32_t*, const int32_t*, int64_t, int, int, int64_t,
                                            int64_t, int64_t, int64_t, int64_t, int64_t, int, float, hipStream_t);

}  // namespace semipd
