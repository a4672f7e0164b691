// Decode-instance dense layers on a CU share: out[m, n] = sum_k x[m, k] * W[n, k], m <= 128 rows.
//
// A decode step reads every weight once (16 GB for Llama-3-8B) and is bound by that stream, on whatever part
// of the chip the decode instance owns.  The library GEMMs launch grids shaped for the whole device (two
// rounds under a CU mask) and never run below ~19 us; this kernel launches about one workgroup per CU of a
// half-chip share, needs no particular CU count to be efficient (a workgroup streams its own rows, nothing
// is persistent) and moves the weights with LDS-DMA (see the block comment at the kernel).  Epilogues: plain store, or
// SiLU(gate) * up for a merged gate_up weight (a wave owns 16 gate rows and the 16 up rows that belong to
// them), which saves the activation kernel and the [m, 2 x inter] round trip.
// C^T[n, m] tiles of v_mfma_f32_16x16x32: a lane ends with 4 consecutive n of one row m (8-byte stores).
// replaces UnquantizedLinearMethod.apply -> F.linear (+ SiluAndMul) at decode batch sizes
//   (layers/linear.py:165-172, models/llama.py:88-92, layers/activation.py:41-53).
#include "common.h"

namespace semipd {

typedef float sl_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sl_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sl_f16x8 __attribute__((ext_vector_type(8)));

union SlFrag {
  uint4 u;
  sl_bf16x8 b;
  sl_f16x8 f;
};
template <typename T> struct SlMfma;
template <> struct SlMfma<bf16_t> {
  __device__ static inline sl_f32x4 mma(const SlFrag& a, const SlFrag& b, sl_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, c, 0, 0, 0);
  }
};
template <> struct SlMfma<f16_t> {
  __device__ static inline sl_f32x4 mma(const SlFrag& a, const SlFrag& b, sl_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.f, b.f, c, 0, 0, 0);
  }
};

typedef uint32_t sl_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint4 sl_load_nt(const void* p) {  // streamed once: keep it out of the way of x in L2 / MALL
  const sl_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const sl_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

enum { SL_PLAIN = 0, SL_SILU_MUL = 1 };

// Weights through LDS-DMA.  A fragment-shaped register load (16 rows x 64 B per wave instruction) costs the
// CU's texture-address path ~80 cycles however it is scheduled: a first version of this file that streamed
// weight AND activation fragments straight into registers topped out at ~17 GB/s of weights per CU
// (profiles/r02_kbench_stream_linear_v1_register_fragments.txt).  global_load_lds moves 1 KB per wave
// instruction as four contiguous 256-byte row segments straight into LDS, costs no VGPRs while in flight and
// leaves the VALU / MFMA free, so the in-flight window per CU is bounded by LDS, not by registers:
//   * every wave owns a private ring of R slots (slot = NG x 16 weight rows x 128 k = NG x 4 KB) and keeps R - 1
//     blocks in flight; the activation block of the same 128 k (MT x 4 KB) is loaded once per WORKGROUP into a
//     shared ring of the same depth, each wave issuing its share -- so the only synchronisation is one raw
//     s_barrier per 128 k, and counted s_waitcnt vmcnt(N) (in order: x share and W of a block are issued
//     together) never drains the stream;
//   * LDS images are lane-linear (that is what the DMA writes), so the bank-conflict swizzle is applied to the
//     SOURCE address: 16-byte chunk c of row i sits at position c ^ (i & 15) of its 256-byte LDS row, and the
//     ds_read_b128 of an MFMA fragment reads (4 s + kq) ^ i -- 16 distinct slots per 16-lane group;
//   * workgroup = NW waves x (NG x 16) weight rows over one K slice; the launch is n_row_batches x KS
//     workgroups, KS chosen from the shape so that a half-chip share is filled by one round; KS > 1 writes fp32
//     planes [KS][M][N] that splitk_planes_reduce below sums in slice order.
template <int MT, int NG, int NW, int R, int RX = R>
struct SgLayout {
  static constexpr int kXStage = MT * 4096;                 // one activation block: MT x 16 rows x 256 B
  static constexpr int kWSlot = NG * 4096;                  // one weight block of a wave
  static constexpr int kXRing = RX * kXStage;               // (RX < R: the wide form, see the kernel's main loop)
  static constexpr int kBytes = kXRing + NW * R * kWSlot;
  static constexpr int kNX = (4 * MT + NW - 1) / NW;        // x DMA pieces per wave per block
  static constexpr int kPerBlock = kNX + 4 * NG;            // VMEM ops a wave issues per block
};

#define SG_GLDS(gp, lp, aux) \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp), \
                                   (__attribute__((address_space(3))) void*)(lp), 16, 0, aux)

template <int N> __device__ inline void sg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Fragment reads are inline asm: a ds_read the compiler can see makes it wait vmcnt(0) for every LDS-DMA in flight
// (it cannot tell the ring slots apart), which would drain the weight stream at every 128 k.  The asm reads are
// ordered by hand: in-order lgkmcnt, a scheduling barrier between the wait and the MFMAs that consume them.
__device__ inline uint4 sg_lds_read16(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int N> __device__ inline void sg_wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ inline uint32_t sg_lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// Grouped form (GR; the expert GEMMs of a decode-sized fused MoE, fused_moe.py:54-273): blockIdx.y = a block of 16 MT
// entries of sorted_token_ids (moe_align_block_size with block size 16 MT: one expert per block, expert_ids[block]); entry
// id reads activation row id / top_k_div and writes output row id (entries >= num_valid are padding), optionally
// scaled by topk_weights[id]; W = w[expert]; no K split.
struct SgGroup {
  const int32_t* sorted_ids;
  const int32_t* expert_ids;
  const int32_t* num_post_pad;
  const float* topk_weights;
  int num_valid, top_k_div, mul_routed_weight;
};

template <typename T, int MT, int NG, int NW, int R, int EPI, bool GR = false, bool ROT = true, int RX = R>
__global__ void __launch_bounds__(64 * NW)
stream_gemm_glds_kernel(T* __restrict__ out, float* __restrict__ planes, const T* __restrict__ x,
                        const T* __restrict__ w, int M, int N, int K, int64_t ldx, int64_t ldo, int kb_per_slice,
                        int planes_only, SgGroup grp = SgGroup()) {
  using L = SgLayout<MT, NG, NW, R, RX>;
  static_assert(RX == R || (RX == 2 && R == 3), "activation ring: as deep as the weight rings, or two slots under three");
  extern __shared__ __attribute__((aligned(16))) char sg_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int n_half = N >> 1;
  const int ks = GR ? 0 : blockIdx.y;
  const int nkb_total = K >> 7;
  const int kb0 = ks * kb_per_slice;
  const int nkb = min(kb_per_slice, nkb_total - kb0);       // >= 1 by construction of the grid
  const int m_blk = GR ? (int)blockIdx.y * (16 * MT) : 0;   // first entry of this block in sorted_token_ids
  if (GR) {
    if (m_blk >= grp.num_post_pad[0]) return;               // whole workgroup: blocks past the padded token count
    w += (int64_t)grp.expert_ids[blockIdx.y] * N * K;
  }
  // ---- DMA source pointers (per lane): piece j = rows 4j .. 4j+3 of a 16-row block, lane -> row 4j + (lane >> 4),
  //      chunk (lane & 15) ^ (row & 15) of the row's 256-byte k-block segment ----
  const int prow = lane >> 4;                               // row inside a piece
  const int rows_per_wg = (EPI == SL_SILU_MUL ? 16 : 16 * NG) * NW;
  const int n0 = blockIdx.x * rows_per_wg + wave * (EPI == SL_SILU_MUL ? 16 : 16 * NG);
  const T* wsrc[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * j + prow;
      const int row = (EPI == SL_SILU_MUL ? g * n_half + n0 : n0 + g * 16) + i;
      wsrc[g][j] = w + (int64_t)min(row, N - 1) * K + (int64_t)kb0 * 128 + (((lane & 15) ^ (i & 15)) << 3);
    }
  const T* xsrc[L::kNX];
  int xdst[L::kNX];
#pragma unroll
  for (int e = 0; e < L::kNX; ++e) {
    const int piece = (wave + e * NW) % (4 * MT);           // duplicates (same bytes, same place) when 4 MT % NW != 0
    const int t = piece >> 2, j = piece & 3;
    const int i = 4 * j + prow;
    int xrow = t * 16 + i;
    if (GR) {
      const int id = grp.sorted_ids[m_blk + xrow];
      xrow = id < grp.num_valid ? id / grp.top_k_div : 0;   // padding entries read row 0 (discarded)
    } else {
      xrow = min(xrow, M - 1);
    }
    xsrc[e] = x + (int64_t)xrow * ldx + (int64_t)kb0 * 128 + (((lane & 15) ^ (i & 15)) << 3);
    xdst[e] = t * 4096 + j * 1024;
  }
  char* const xring = sg_smem;
  char* const wring = sg_smem + L::kXRing + wave * (R * L::kWSlot);
  const uint32_t xring_addr = sg_lds_addr(xring), wring_addr = sg_lds_addr(wring);

  // Workgroups do not walk K in lock-step: workgroup i starts at block rot(i) of its slice and wraps around.  Every
  // row of W starts at the same offset modulo the row pitch (8 KB for K = 4096), so workgroups that all read k-block b
  // of their rows at the same moment put their whole load on the few HBM channels that hold byte range b of a pitch:
  // on a 48-CU share the kernel was pinned at 1.35 TB/s however many of the CUs streamed
  // (profiles/r05_kbench_stream_planes_graph_48cus_v0.txt).  The rotation changes the order of a workgroup's fp32
  // sums over k (a function of blockIdx only: the same bits on every run, grid and CU mask).
  // (not in the grouped form: its experts' fused and plain launches promise equal bits, tests/test_gpu_ops.py
  //  test_moe_stream_gemm_matches_fp32_per_expert, and its workgroups start at different experts' weights anyway)
  const int rot = (ROT && !GR) ? (int)(((uint32_t)(blockIdx.x + 1) * 0x9E3779B1u >> 16) % (uint32_t)nkb) : 0;
  auto issue_x = [&](int kb) __attribute__((always_inline)) {  // activation block kb of this slice -> ring position kb % RX
    int kk = kb + rot;
    if (kk >= nkb) kk -= nkb;
    const int64_t koff = (int64_t)kk * 128;
#pragma unroll
    for (int e = 0; e < L::kNX; ++e) SG_GLDS(xsrc[e] + koff, xring + (kb % RX) * L::kXStage + xdst[e], 0);
  };
  auto issue_w = [&](int kb) __attribute__((always_inline)) {  // this wave's weight block kb -> ring position kb % R
    int kk = kb + rot;
    if (kk >= nkb) kk -= nkb;
    const int64_t koff = (int64_t)kk * 128;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) SG_GLDS(wsrc[g][j] + koff, wring + (kb % R) * L::kWSlot + g * 4096 + j * 1024, 2);
  };
  auto issue = [&](int kb) __attribute__((always_inline)) {
    issue_x(kb);
    issue_w(kb);
  };

  sl_f32x4 acc[NG][MT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[g][t] = sl_f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offset of this lane inside a 16-row x 256-byte image, k-step s: row c16, chunk (4 s + q4) ^ c16
  int frag_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) frag_off[s] = c16 * 256 + (((4 * s + q4) ^ c16) << 4);

  if (RX == R) {
#pragma unroll
    for (int p = 0; p < R - 1; ++p)
      if (p < nkb) issue(p);
  } else {
    // 65 .. 128 rows: an activation block is 24 / 32 KB and three slots of it do not fit next to the weight rings, but the
    // activations come from L2 and the weights from HBM: TWO weight blocks stay in flight per wave, ONE activation block.
    // Issue order w(0) x(0) w(1), then per block x(kb + 1) w(kb + 2): the counted wait below leaves exactly the youngest
    // weight block outstanding.
    issue_w(0);
    issue_x(0);
    if (1 < nkb) issue_w(1);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    // block kb landed (this wave's share); the R - 2 younger blocks stay in flight.  Near the end of the slice
    // fewer blocks are outstanding than the count assumes: drain.
    if (RX == R) {
      if (kb + R - 1 <= nkb) sg_wait_vm<(R - 2) * L::kPerBlock>();
      else sg_wait_vm<0>();
    } else {
      if (kb + 1 < nkb) sg_wait_vm<4 * NG>();
      else sg_wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();       // every wave's share of x(kb) landed; everybody is done with block kb - 1
    asm volatile("" ::: "memory");
    if (RX == R) {
      if (kb + R - 1 < nkb) issue(kb + R - 1);              // into the ring position block kb - 1 just left
    } else {
      if (kb + 1 < nkb) issue_x(kb + 1);                    // both into the positions block kb - 1 just left
      if (kb + 2 < nkb) issue_w(kb + 2);
    }
    const uint32_t xs = xring_addr + (kb % RX) * L::kXStage, wsl = wring_addr + (kb % R) * L::kWSlot;
    SlFrag a[2][NG], b[2][MT];
    auto read_step = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) a[buf][g].u = sg_lds_read16(wsl + g * 4096 + frag_off[s]);
#pragma unroll
      for (int t = 0; t < MT; ++t) b[buf][t].u = sg_lds_read16(xs + t * 4096 + frag_off[s]);
    };
    read_step(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
        read_step((s + 1) & 1, s + 1);
        sg_wait_lgkm<NG + MT>();          // the fragments of k-step s are in; those of s + 1 are on their way
      } else {
        sg_wait_lgkm<0>();
      }
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g][t] = SlMfma<T>::mma(a[s & 1][g], b[s & 1][t], acc[g][t]);
    }
  }

  // ---- epilogue: lane holds C[m = t*16 + c16][n = nbase + q4*4 + r] ----
  if (!GR && (gridDim.y > 1 || planes_only)) {
    float* pl = planes + (int64_t)ks * M * N;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + c16;
        const int n = (EPI == SL_SILU_MUL ? g * n_half + n0 : n0 + g * 16) + q4 * 4;
        if (m < M && n < N) *reinterpret_cast<sl_f32x4*>(pl + (int64_t)m * N + n) = acc[g][t];
      }
    return;
  }
  if (EPI == SL_PLAIN) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c16;
        const int n = n0 + g * 16 + q4 * 4;
        float scale = 1.f;
        if (GR) {
          m = grp.sorted_ids[m_blk + m];                    // output row = the routed entry itself
          if (m < grp.num_valid && grp.mul_routed_weight) scale = grp.topk_weights[m];
        }
        if (m < M && n < N) {
          sl_f32x4 v = acc[g][t];
          if (GR) v *= scale;
          uint2 p;
          p.x = (uint32_t)Elem<T>::from_f(v[0]).v | ((uint32_t)Elem<T>::from_f(v[1]).v << 16);
          p.y = (uint32_t)Elem<T>::from_f(v[2]).v | ((uint32_t)Elem<T>::from_f(v[3]).v << 16);
          *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = p;
        }
      }
  } else {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      int m = t * 16 + c16;
      const int n = n0 + q4 * 4;
      if (GR) m = grp.sorted_ids[m_blk + m];
      if (m < M && n < n_half) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the unfused path rounds the GEMM output to T before the activation reads it
          const float gq = Elem<T>::to_f(Elem<T>::from_f(acc[0][t][j])), uq = Elem<T>::to_f(Elem<T>::from_f(acc[NG - 1][t][j]));
          r[j] = gq / (1.f + __expf(-gq)) * uq;
          // keep the product an fp32 VALUE: left alone the compiler folds multiply + conversion into one
          // v_fma_mixlo_f16 (a single rounding), one ulp away from silu_and_mul's two in rare cases
          asm volatile("" : "+v"(r[j]));
        }
        uint2 p;
        p.x = (uint32_t)Elem<T>::from_f(r[0]).v | ((uint32_t)Elem<T>::from_f(r[1]).v << 16);
        p.y = (uint32_t)Elem<T>::from_f(r[2]).v | ((uint32_t)Elem<T>::from_f(r[3]).v << 16);
        *reinterpret_cast<uint2*>(out + (int64_t)m * ldo + n) = p;
      }
    }
  }
}

// out[m, n] = T(sum_z planes[z][m][n]) in slice order; SiLU * mul variant reads gate / up columns n, n + N/2
template <typename T, int EPI>
__global__ void __launch_bounds__(256)
splitk_planes_reduce_kernel(T* __restrict__ out, const float* __restrict__ planes, int ksplit, int M, int N, int64_t ldo) {
  const int n_out = EPI == SL_SILU_MUL ? N / 2 : N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = n_out / 4;
  if (i >= (int64_t)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
  const int64_t plane = (int64_t)M * N;
  const float* src = planes + (int64_t)m * N + n;
  // every plane's load in flight at once (common.h: planes_sum_f4), slice order
  float4 s4[EPI == SL_SILU_MUL ? 2 : 1];
  if (EPI == SL_SILU_MUL) {
    const float* const pp[2] = {src, src + N / 2};
    float4 t2[2];
    planes_sum_f4<2>(pp, ksplit, plane, t2);
    s4[0] = t2[0];
    s4[EPI == SL_SILU_MUL ? 1 : 0] = t2[1];
  } else {
    const float* const pp[1] = {src};
    float4 t1[1];
    planes_sum_f4<1>(pp, ksplit, plane, t1);
    s4[0] = t1[0];
  }
  float r[4] = {s4[0].x, s4[0].y, s4[0].z, s4[0].w};
  if (EPI == SL_SILU_MUL) {
    const float4 u4 = s4[EPI == SL_SILU_MUL ? 1 : 0];
    const float u[4] = {u4.x, u4.y, u4.z, u4.w};
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

// K slices of a launch with n_rb row batches on a share of `cus` compute units (one 8-wave workgroup per CU: the LDS
// rings take the whole CU).  The hardware runs the n_rb x ks workgroups in ceil(n_rb * ks / cus) rounds, so the split is
// picked to fill whole rounds of THIS share: a cost model in microseconds -- a workgroup streams one 128-k block of
// its rows (NW x NG x 4 KB) in ~0.86 us (37 GB/s per CU, profiles/r03_kbench_stream_linear_small_shares_v0.txt), a
// round costs ~2 us of ramp and drain, and every slice beyond the first writes and re-reads an fp32 plane.  With the
// default share (128 CUs, a half-chip decode instance) this reproduces the round-2 table (o_proj / down 4, qkv 2,
// gate_up 1).  The split -- and with it the summation order, i.e. the last bit of the result -- depends on the share the
// process declares (semipd_stream_linear_set_cus); two processes that must produce the same bits declare the same.
static std::atomic<int> g_sl_cus{128};

static int sg_pick_ksplit(int n_rb, int nkb, int M, int N, bool extra_reduce_launch, int wgs_per_cu = 1) {
  const int cus = max(8, g_sl_cus.load(std::memory_order_relaxed)) * wgs_per_cu;
  const float t_block = 0.86f, t_round = 2.0f;
  const float plane_us = 2.f * M * (float)N * 4.f / (cus / wgs_per_cu * 37e3f);   // one plane written + read back
  int best = 1;
  float best_cost = 1e30f;
  for (int ksp = 1; ksp <= 16; ++ksp) {
    const int per = (nkb + ksp - 1) / ksp;
    if (ksp > 1 && per < 4) break;
    const int eff = (nkb + per - 1) / per;                         // no empty slices
    if (eff != ksp) continue;
    const int rounds = (n_rb * eff + cus - 1) / cus;
    float cost = rounds * (per * t_block + t_round);
    // every slice beyond the first: its plane's round trip + ~1 us in the consumer (whole chip, o_proj: 8 slices 15.3 us,
    // 4 slices 13.1 us, profiles/r03_kbench_stream_linear_whole_chip.txt); a separate reduction launch: ~5 us + its ramp
    if (eff > 1) cost += eff * (plane_us + 1.0f) + (extra_reduce_launch ? 8.f : 0.f);
    if (cost < best_cost - 1e-3f) best_cost = cost, best = eff;
  }
  return best;
}

template <typename T, int MT, int NG, int NW, int R, int EPI, int RX = R>
static int sg_launch(T* out, float* planes, size_t planes_bytes, const T* x, const T* w, int M, int N, int K,
                     int64_t ldx, int64_t ldo, int force_ks, hipStream_t st, int* planes_only_ks = nullptr) {
  using L = SgLayout<MT, NG, NW, R, RX>;
  const int rows_per_wg = (EPI == SL_SILU_MUL ? 16 : 16 * NG) * NW;
  const int n_rows = EPI == SL_SILU_MUL ? N / 2 : N;
  const int n_rb = (n_rows + rows_per_wg - 1) / rows_per_wg;
  const int nkb = K / 128;
  // (the separate-reduction charge applies to the SiLU epilogue only: a plain layer's planes are usually summed by its
  //  consumer, and its reducing and plane-returning forms must pick the SAME split to produce the same bits)
  // (four-wave workgroups of the plain epilogue at <= 32 rows: 72 KB of LDS, two of them share a CU)
  const int wgs_per_cu = (EPI == SL_PLAIN && NW == 4 && NG == 1 && RX == R && L::kBytes * 2 <= 160 * 1024) ? 2 : 1;
  int ksp = force_ks > 0 ? force_ks : sg_pick_ksplit(n_rb, nkb, M, N, EPI == SL_SILU_MUL, wgs_per_cu);
  while (ksp > 1 && (size_t)ksp * M * N * 4 > planes_bytes) --ksp;
  const int per = (nkb + ksp - 1) / ksp;
  ksp = (nkb + per - 1) / per;
  static std::atomic<uint64_t> lds_ok{0};  // one per instantiation, one bit per device
  static const bool rot_off = [] { const char* e = getenv("SEMIPD_SL_ROT"); return e && atoi(e) == 0; }();   // A/B knob
  if (rot_off) {
    static std::atomic<uint64_t> lds_ok0{0};
    if (ensure_dynamic_lds((const void*)stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, false, false, RX>, L::kBytes, lds_ok0, "stream_gemm_glds"))
      return 1;
    hipLaunchKernelGGL((stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, false, false, RX>), dim3(n_rb, ksp), dim3(64 * NW), L::kBytes, st,
                       out, planes, x, w, M, N, K, ldx, ldo, per, planes_only_ks ? 1 : 0);
  } else {
  if (ensure_dynamic_lds((const void*)stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, false, true, RX>, L::kBytes, lds_ok, "stream_gemm_glds"))
    return 1;
  hipLaunchKernelGGL((stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, false, true, RX>), dim3(n_rb, ksp), dim3(64 * NW), L::kBytes, st,
                     out, planes, x, w, M, N, K, ldx, ldo, per, planes_only_ks ? 1 : 0);
  }
  int rc = launch_status("stream_gemm_glds");
  if (planes_only_ks) {   // the consumer sums the planes (e.g. semipd_fused_add_rmsnorm_planes)
    *planes_only_ks = ksp;
    return rc;
  }
  if (rc || ksp == 1) return rc;
  const int n_out = EPI == SL_SILU_MUL ? N / 2 : N;
  const int64_t items = (int64_t)M * (n_out / 4);
  hipLaunchKernelGGL((splitk_planes_reduce_kernel<T, EPI>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, out,
                     (const float*)planes, ksp, M, N, ldo);
  return launch_status("splitk_planes_reduce");
}

template <typename T, int MT, int EPI>
static int sg_launch_grouped(T* c, const T* a, const T* w, const SgGroup& grp, int64_t max_sorted, int N, int K, int64_t lda,
                             int64_t ldc, hipStream_t st) {
  constexpr int NG = EPI == SL_SILU_MUL ? 2 : 1, NW = EPI == SL_SILU_MUL ? 4 : 8, R = 3;
  using L = SgLayout<MT, NG, NW, R>;
  const int rows_per_wg = (EPI == SL_SILU_MUL ? 16 : 16 * NG) * NW;
  const int n_rows = EPI == SL_SILU_MUL ? N / 2 : N;
  const int n_rb = (n_rows + rows_per_wg - 1) / rows_per_wg;
  const int blocks = (int)(max_sorted / (16 * MT));
  if (blocks == 0) return 0;
  static std::atomic<uint64_t> lds_ok{0};
  if (ensure_dynamic_lds((const void*)stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, true>, L::kBytes, lds_ok,
                         "stream_gemm_glds_grouped"))
    return 1;
  hipLaunchKernelGGL((stream_gemm_glds_kernel<T, MT, NG, NW, R, EPI, true>), dim3(n_rb, blocks), dim3(64 * NW), L::kBytes, st,
                     c, (float*)nullptr, a, w, grp.num_valid, N, K, lda, ldc, K / 128, 0, grp);
  return launch_status("stream_gemm_glds_grouped");
}

}  // namespace semipd

using namespace semipd;

extern "C" {

// tuning knob for tools/kbench.py (not part of the ABI): SEMIPD_SL_KS forces the number of K slices
static int sl_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

/* CUs of the share this process runs its decode-sized GEMMs on (HSA_CU_MASK / stream mask); 0 restores the default
 * (128).  Picks the K split of semipd_stream_linear / _planes, see sg_pick_ksplit. */
int semipd_stream_linear_set_cus(int cus) {
  SEMIPD_CHECK_ARG(cus >= 0 && cus <= 4096, SEMIPD_EINVAL, "stream_linear_set_cus: bad CU count %d", cus);
  g_sl_cus.store(cus == 0 ? 128 : cus, std::memory_order_relaxed);
  owned_cus().store(cus, std::memory_order_relaxed);
  return 0;
}

// Narrow workgroups (four waves = 64 weight rows; 72 KB of LDS up to 32 rows: two per CU) for the plain epilogue of a SMALL
// weight at <= 64 rows: o_proj of Llama-3-8B (4096 x 4096) is 32 workgroups of 128 rows per K slice -- with the four slices the cost rule
// picks, a workgroup on half of the chip's CUs, and what a CU pulls from HBM is the limit this kernel runs into.  Measured
// (profiles/r06_kbench_narrow_workgroups.txt, 32 rows, whole chip): o_proj 10.4 -> 8.5 us, the 70B TP = 8 rank's qkv (1280 x
// 8192) 10.0 -> 7.7, its o_proj (8192 x 1024) 7.2 -> 6.1; at 64 rows 11.8 -> 9.6, 11.7 -> 9.0, 8.4 -> 6.9; no gain from 25 M
// weight elements up (qkv 8B, down_proj).
// SEMIPD_SL_NW: 4 = always where it applies, 8 = never; otherwise weights of at most SEMIPD_SL_NARROW_MAX_NK elements (2^24).
static bool sg_narrow(int mt, int N, int K, int fuse_silu_mul) {
  if (fuse_silu_mul || mt > 4) return false;   // (33 .. 64 rows: 84 / 96 KB of LDS, one workgroup per CU -- still a workgroup on every CU)
  const int knob = sl_env("SEMIPD_SL_NW", 0);
  if (knob == 4) return true;
  if (knob == 8) return false;
  static const int64_t max_nk = [] { const char* e = getenv("SEMIPD_SL_NARROW_MAX_NK"); return e ? atoll(e) : (1ll << 24); }();
  return (int64_t)N * K <= max_nk;
}

// The same idea at 65 .. 128 rows: one group of 16 weight rows per wave (64 rows per workgroup, 112 KB of LDS) instead of two.
// Measured (profiles/r06_kbench_wide_rows_stream_vs_tiled.txt, second table; 80 / 128 rows): o_proj 17.5 -> 14.6 / 21.8 ->
// 18.4 us, qkv (25 M elements) 20.2 -> 19.4 / 25.1 -> 23.1, down_proj (59 M) 29.8 -> 28.4 / 35.4 -> 36.9: weights of up to
// 2^25 elements.  SEMIPD_SL_WIDE_NARROW: 2 = always, 0 = never.
static bool sg_wide_narrow(int N, int K, int fuse_silu_mul) {
  if (fuse_silu_mul) return false;
  const int knob = sl_env("SEMIPD_SL_WIDE_NARROW", 1);
  if (knob == 2) return true;
  return knob == 1 && (int64_t)N * K <= (1ll << 25);
}

size_t semipd_stream_linear_workspace(int64_t max_n) {
  return (size_t)16 * 64 * (size_t)max_n * 4;   // 16 K slices of [64 rows, n] fp32
}

int semipd_stream_linear(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes,
                         int64_t rows, int64_t n, int64_t k, int64_t ldx, int64_t ldo, int fuse_silu_mul, int dtype,
                         void* stream) {
  SEMIPD_CHECK_ARG(rows >= 0 && n > 0 && k > 0 && ldx >= k, SEMIPD_EINVAL, "stream_linear: bad sizes");
  if (rows == 0) return 0;
  SEMIPD_CHECK_ARG(out && x && weight, SEMIPD_EINVAL, "stream_linear: null pointer");
  SEMIPD_CHECK_ARG(rows <= 128, SEMIPD_ESHAPE, "stream_linear: %lld rows; this is the weight-streaming path for decode "
                   "batches of at most 128 rows", (long long)rows);
  const int64_t n_out = fuse_silu_mul ? n / 2 : n;
  SEMIPD_CHECK_ARG(ldo >= n_out && (!fuse_silu_mul || n % 2 == 0), SEMIPD_EINVAL, "stream_linear: ldo < output width");
  SEMIPD_CHECK_ARG(k % 128 == 0 && ldx % 8 == 0 && n_out % 16 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(weight) &&
                   (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && n < (1 << 30) && k < (1 << 30) &&
                   (!workspace || aligned16(workspace)),
                   SEMIPD_EALIGN, "stream_linear: k %% 128, output width %% 16, 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  const int M = (int)rows, N = (int)n, K = (int)k;
  const int mt = (M + 15) / 16;
  const int force_ks = workspace ? sl_env("SEMIPD_SL_KS", 0) : 1;
  float* planes = (float*)workspace;
  const size_t pb = workspace ? workspace_bytes : 0;
  int rc = 0;
  // ring depth: up to 32 rows the activation ring is small enough for FOUR slots in the CU's 160 KB (three blocks in
  // flight per wave instead of two).  Measured: no gain on 64 / 96 / 128-CU shares (profiles/
  // r03_kbench_stream_linear_ring4_vs_ring3.txt) -- a CU pulls ~35-41 GB/s from HBM however deep its rings are -- so the
  // default stays three slots; SEMIPD_SL_RING=4 selects the deep instantiation
  static const int ring_knob = sl_env("SEMIPD_SL_RING", 3);
  const bool deep = mt <= 2 && ring_knob >= 4;
#define SL_GO(MTV, RV) \
  if (fuse_silu_mul) { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 2, 4, RV, SL_SILU_MUL>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st))); } \
  else { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 8, RV, SL_PLAIN>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st))); }
  // 65 .. 128 rows (SL_WIDE): two weight-row groups per wave (an activation fragment feeds two MFMAs: the LDS reads per
  // weight byte of MT = 8 with one group would be the bound), four waves, weight rings of three slots and an activation
  // ring of TWO -- a block of 128 rows is 32 KB: 64 + 96 KB is the CU's LDS to the byte
#define SL_WIDE(MTV) \
  if (fuse_silu_mul) { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 2, 4, 3, SL_SILU_MUL, 2>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st))); } \
  else { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 2, 4, 3, SL_PLAIN, 2>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st))); }
  if (sg_narrow(mt, N, K, fuse_silu_mul)) {
#define SL_NARROW(MTV) SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 4, 3, SL_PLAIN>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st)));
    if (mt == 1) { SL_NARROW(1) } else if (mt == 2) { SL_NARROW(2) } else if (mt == 3) { SL_NARROW(3) } else { SL_NARROW(4) }
#undef SL_NARROW
  }
  else if (mt == 1) { if (deep) { SL_GO(1, 4) } else { SL_GO(1, 3) } }
  else if (mt == 2) { if (deep) { SL_GO(2, 4) } else { SL_GO(2, 3) } }
  else if (mt == 3) { SL_GO(3, 3) } else if (mt == 4) { SL_GO(4, 3) }
  else if (sg_wide_narrow(N, K, fuse_silu_mul)) {
#define SL_WN(MTV) SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 4, 3, SL_PLAIN, 2>((T*)out, planes, pb, (const T*)x, (const T*)weight, M, N, K, ldx, ldo, force_ks, st)));
    if (mt <= 6) { SL_WN(6) } else { SL_WN(8) }
#undef SL_WN
  }
  else if (mt <= 6) { SL_WIDE(6) } else { SL_WIDE(8) }
#undef SL_WIDE
#undef SL_GO
  return rc;
}

/* The same GEMM, stopped before the reduction: fp32 planes [*ksplit][rows][n] in `planes` (at least
 * semipd_stream_linear_workspace bytes), for a consumer that sums them itself (semipd_fused_add_rmsnorm_planes). */
int semipd_stream_linear_planes(float* planes, size_t planes_bytes, const void* x, const void* weight, int64_t rows,
                                int64_t n, int64_t k, int64_t ldx, int dtype, int* ksplit, void* stream) {
  SEMIPD_CHECK_ARG(rows > 0 && rows <= 128 && n > 0 && k > 0 && ldx >= k && ksplit, SEMIPD_EINVAL,
                   "stream_linear_planes: bad sizes");
  SEMIPD_CHECK_ARG(planes && x && weight, SEMIPD_EINVAL, "stream_linear_planes: null pointer");
  SEMIPD_CHECK_ARG(k % 128 == 0 && ldx % 8 == 0 && n % 16 == 0 && aligned16(x) && aligned16(weight) && aligned16(planes) &&
                   n < (1 << 30) && k < (1 << 30) && planes_bytes >= (size_t)rows * n * 4,
                   SEMIPD_EALIGN, "stream_linear_planes: k %% 128, n %% 16, 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  const int M = (int)rows, N = (int)n, K = (int)k;
  const int mt = (M + 15) / 16;
  int rc = 0;
  static const int ring_knob = sl_env("SEMIPD_SL_RING", 3);
  const bool deep = mt <= 2 && ring_knob >= 4;
#define SL_GO(MTV, RV) \
  SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 8, RV, SL_PLAIN>((T*)nullptr, planes, planes_bytes, (const T*)x, (const T*)weight, M, N, K, ldx, N, sl_env("SEMIPD_SL_KS", 0), st, ksplit)));
#define SL_WIDE(MTV) \
  SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 2, 4, 3, SL_PLAIN, 2>((T*)nullptr, planes, planes_bytes, (const T*)x, (const T*)weight, M, N, K, ldx, N, sl_env("SEMIPD_SL_KS", 0), st, ksplit)));
  if (sg_narrow(mt, N, K, 0)) {
#define SL_NARROW(MTV) SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 4, 3, SL_PLAIN>((T*)nullptr, planes, planes_bytes, (const T*)x, (const T*)weight, M, N, K, ldx, N, sl_env("SEMIPD_SL_KS", 0), st, ksplit)));
    if (mt == 1) { SL_NARROW(1) } else if (mt == 2) { SL_NARROW(2) } else if (mt == 3) { SL_NARROW(3) } else { SL_NARROW(4) }
#undef SL_NARROW
  }
  else if (mt == 1) { if (deep) { SL_GO(1, 4) } else { SL_GO(1, 3) } }
  else if (mt == 2) { if (deep) { SL_GO(2, 4) } else { SL_GO(2, 3) } }
  else if (mt == 3) { SL_GO(3, 3) } else if (mt == 4) { SL_GO(4, 3) }
  else if (sg_wide_narrow(N, K, 0)) {
#define SL_WN(MTV) SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 4, 3, SL_PLAIN, 2>((T*)nullptr, planes, planes_bytes, (const T*)x, (const T*)weight, M, N, K, ldx, N, sl_env("SEMIPD_SL_KS", 0), st, ksplit)));
    if (mt <= 6) { SL_WN(6) } else { SL_WN(8) }
#undef SL_WN
  }
  else if (mt <= 6) { SL_WIDE(6) } else { SL_WIDE(8) }
#undef SL_WIDE
#undef SL_GO
  return rc;
}

/* out_f32[rows, n] = x[rows, k] @ weight[n, k]^T in fp32 (no K split: the accumulators are stored as they are): the
 * logits GEMM of a decode batch (layers/logits_processor.py:394-445 computes them in the activation type and converts;
 * here the fp32 sums reach the sampler unrounded).  rows <= 64, k % 128 == 0, n % 16 == 0. */
int semipd_stream_linear_f32(float* out, const void* x, const void* weight, int64_t rows, int64_t n, int64_t k, int64_t ldx,
                             int dtype, void* stream) {
  SEMIPD_CHECK_ARG(rows > 0 && rows <= 64 && n > 0 && k > 0 && ldx >= k, SEMIPD_EINVAL, "stream_linear_f32: bad sizes");
  SEMIPD_CHECK_ARG(out && x && weight, SEMIPD_EINVAL, "stream_linear_f32: null pointer");
  SEMIPD_CHECK_ARG(k % 128 == 0 && ldx % 8 == 0 && n % 16 == 0 && aligned16(x) && aligned16(weight) && aligned16(out) &&
                   n < (1 << 30) && k < (1 << 30),
                   SEMIPD_EALIGN, "stream_linear_f32: k %% 128, n %% 16, 16-byte aligned rows required");
  hipStream_t st = as_stream(stream);
  const int M = (int)rows, N = (int)n, K = (int)k;
  const int mt = (M + 15) / 16;
  static const int ring_knob = sl_env("SEMIPD_SL_RING", 3);
  const bool deep = mt <= 2 && ring_knob >= 4;
  int rc = 0, ks = 0;
#define SL_GO(MTV, RV) \
  SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch<T, MTV, 1, 8, RV, SL_PLAIN>((T*)nullptr, out, (size_t)M * N * 4, (const T*)x, (const T*)weight, M, N, K, ldx, N, 1, st, &ks)));
  if (mt == 1) { if (deep) { SL_GO(1, 4) } else { SL_GO(1, 3) } }
  else if (mt == 2) { if (deep) { SL_GO(2, 4) } else { SL_GO(2, 3) } }
  else if (mt == 3) { SL_GO(3, 3) } else { SL_GO(4, 3) }
#undef SL_GO
  return rc;
}

/* invoke_fused_moe_kernel (fused_moe.py:501-612) for DECODE-sized calls with the LDS-DMA streaming kernel: every expert's
 * weights are read once, by row-shaped LDS-DMA (a CU sustains ~36-41 GB/s of those against ~23 GB/s of the register-
 * fragment loads of semipd_moe_grouped_gemm's streaming kernel: what counts on a partial CU share).  sorted_token_ids /
 * expert_ids from moe_align_block_size with block size block_m in {16, 32, 48, 64} (a decode batch of T tokens routes
 * at most T rows to one expert, so block_m = 16 ceil(T / 16) never splits an expert); max_sorted = entries of
 * sorted_token_ids, a multiple of block_m.  c[id, :] = a[id / top_k_div, :] @ w[expert]^T for id < num_valid, times
 * topk_weights[id] when mul_routed_weight; fuse_silu_mul: w[e] = merged [gate; up] and c = SiLU(gate) * up. */
int semipd_moe_stream_gemm(void* c, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
                           const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                           int64_t k, int64_t max_sorted, int top_k_div, int mul_routed_weight, int fuse_silu_mul, int block_m,
                           int dtype, void* stream) {
  SEMIPD_CHECK_ARG(num_valid >= 0 && n > 0 && k > 0 && max_sorted >= 0 && top_k_div > 0, SEMIPD_EINVAL,
                   "moe_stream_gemm: bad sizes");
  if (num_valid == 0 || max_sorted == 0) return 0;
  SEMIPD_CHECK_ARG(c && a && w && sorted_token_ids && expert_ids && num_tokens_post_pad, SEMIPD_EINVAL,
                   "moe_stream_gemm: null pointer");
  SEMIPD_CHECK_ARG(!mul_routed_weight || topk_weights, SEMIPD_EINVAL, "moe_stream_gemm: topk_weights required");
  const int64_t n_out = fuse_silu_mul ? n / 2 : n;
  SEMIPD_CHECK_ARG((block_m == 16 || block_m == 32 || block_m == 48 || block_m == 64) && max_sorted % block_m == 0 &&
                       k % 128 == 0 && n_out % 16 == 0 && (!fuse_silu_mul || n % 2 == 0) && aligned16(a) && aligned16(w) &&
                       (reinterpret_cast<uintptr_t>(c) & 7u) == 0 && num_valid < (1 << 30) && n < (1 << 30) && k < (1 << 30),
                   SEMIPD_ESHAPE, "moe_stream_gemm: block_m in {16, 32, 48, 64}, k %% 128, output width %% 16 required");
  SgGroup grp{sorted_token_ids, expert_ids, num_tokens_post_pad, topk_weights, (int)num_valid, top_k_div, mul_routed_weight};
  hipStream_t st = as_stream(stream);
  int rc = 0;
#define SGG(MTV) \
  if (fuse_silu_mul) { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch_grouped<T, MTV, SL_SILU_MUL>((T*)c, (const T*)a, (const T*)w, grp, max_sorted, (int)n, (int)k, k, n_out, st))); } \
  else { SEMIPD_DISPATCH_HALF(dtype, T, rc = (sg_launch_grouped<T, MTV, SL_PLAIN>((T*)c, (const T*)a, (const T*)w, grp, max_sorted, (int)n, (int)k, k, n_out, st))); }
  if (block_m == 16) { SGG(1) } else if (block_m == 32) { SGG(2) } else if (block_m == 48) { SGG(3) } else { SGG(4) }
#undef SGG
  return rc;
}

}  // extern "C"
