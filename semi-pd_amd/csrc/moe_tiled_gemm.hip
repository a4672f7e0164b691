// Grouped GEMM of the MoE layers for prefill-sized batches, gfx950.
//
//   C[sorted row r, :N] = A[token(r), :K] @ W[expert(block(r)), :N, :K]^T          (fused_moe_kernel semantics)
//
// The weight-streaming kernel (skinny_gemm.hip) reads an expert's weights once per block of 128 routed rows and
// keeps its MFMA operands in registers: right for decode, where the weights are the traffic, but a prefill chunk
// routes hundreds of rows to every expert and the same kernel tops out at 0.5 PFLOP/s (0.20 of the dense peak;
// profiles/r01_kbench_moe_v5_stream128_ng4.txt).  This one is a tiled GEMM:
//
//   workgroup = 8 waves (2 x 4) on a 128 x 256 output tile, every wave 64 x 64 (2 x 2 MFMA 32x32x16 tiles);
//   A (128 gathered token rows) and W (256 weight rows) arrive k-block by k-block (64 k) through LDS-DMA into a
//   three-deep ring: buffer_load ... lds with per-lane row offsets computed once and the k offset as the scalar
//   operand -- no address arithmetic, no staging registers, no ds_write in the loop; one s_barrier per k-block and a
//   counted vmcnt (the DMA of the block after next stays in flight across it);
//   128 x 256 x 64: 48 KiB of operands per 4.2 MFLOP = 85 FLOP per byte through the CU's 64 B/clk vector memory path
//   (a 128 x 128 tile needs the whole path at the MFMA peak);
//   LDS rows are 128 bytes (64 k), so two rows share a bank line: chunk c of row i sits at position c ^ ((i >> 1) & 7)
//   (applied on the source side of the lane-linear DMA image) and the 16 lanes of a ds_read_b128 group land in 16
//   different 16-byte slots;
//   fragment reads are inline asm with counted lgkmcnt (a ds_read the compiler sees waits vmcnt(0) for the DMA that is
//   in flight on purpose; stream_linear.hip).
//
// Measured (profiles/r02_kbench_moe_tiled_gemm.txt, DeepSeek-V2-Lite experts, 64 x top-6, bf16): GEMM1 / GEMM2 at
// T = 8192 655-729 / 590-683 TFLOP/s against 578 / 549 for the streaming kernel, T = 4096 598 / 565 against 520 / 467.
// With the DMA and the barrier switched off the loop itself runs at 0.99 PFLOP/s: every wave reads 1 KiB of fragments per
// MFMA and the DMA writes another 48 KiB per k-block -- the LDS is busy ~87 % of the matrix pipe's time.  The next step
// is a 128 x 128 register tile per wave (one wave per SIMD, accumulators in AGPRs), which halves the fragment traffic.
//
// Results: fp32 accumulation over k in order, one rounding to T (times the routed weight where asked) -- the same
// contract as the streaming kernel; parity tests are the fused-MoE ones (tests/test_gpu_ops.py, golden/fused_moe.npz).
// replaces fused_moe_kernel / invoke_fused_moe_kernel (layers/moe/fused_moe_triton/fused_moe.py:54-273, 501-612).
#include "common.h"
#include "mfma_frag.h"

#include <type_traits>

namespace semipd {

namespace mtg {

constexpr int kBM = 128, kBN = 256, kBK = 64;
constexpr int kRowB = kBK * 2;                       // bytes of one LDS row
constexpr int kAStage = kBM * kRowB;                 // 16 KiB
constexpr int kBStage = kBN * kRowB;                 // 32 KiB
constexpr int kStage = kAStage + kBStage;            // 48 KiB
constexpr int kRing = 3;
constexpr int kLds = kRing * kStage;                 // 144 KiB
constexpr int kChunk = 4;                             // row blocks per L2-resident chunk (see the tile mapping)
constexpr int kPieces = kStage / 1024 / 8;           // DMA pieces (1 KiB each) per wave and k-block: 6 (2 of A, 4 of W)

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
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace mtg

template <typename T>
__global__ void __launch_bounds__(512, 1)
moe_tiled_gemm_kernel(T* __restrict__ c, const T* __restrict__ a, const T* __restrict__ w,
                      const float* __restrict__ topk_weights, const int32_t* __restrict__ sorted_ids,
                      const int32_t* __restrict__ expert_ids, const int32_t* __restrict__ num_post_pad, int num_valid, int N,
                      int K, int top_k_div, int mul_routed_weight, int a_rows) {
  using namespace mtg;
  extern __shared__ __attribute__((aligned(16))) char mtg_smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)mtg_smem;
  // ---- workgroup -> tile, XCD aware.  Workgroups are dealt to the 8 XCDs round robin by linear id and every XCD has its
  // own 4 MiB L2.  With tiles in plain (x fastest, y) order the row blocks of one expert -- the only workgroups that
  // share a weight tile -- land on different XCDs and each of them pulls the tile from memory again: 12 MB of operands
  // per row block, 5.7 TB/s at 600 TFLOP/s.  Here XCD c owns a contiguous eighth of the row blocks (about E / 8 experts)
  // and walks it in chunks of kChunk row blocks: for each chunk all column tiles, row blocks fastest.  The chunk's
  // token rows (kChunk x 0.5 MiB at K = 2048) stay in that L2 across the columns, and a weight tile is fetched once
  // for the chunk's row blocks of its expert, which run at the same time.
  const int gx = (N + kBN - 1) / kBN;
  const int y_all = __builtin_amdgcn_readfirstlane(*num_post_pad) / kBM;     // row blocks that hold rows
  const int y_per = (y_all + 7) / 8;
  const int xcd = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3;
  const int chunk = s / (kChunk * gx), within = s - chunk * (kChunk * gx);
  const int bx = within / kChunk, by = xcd * y_per + chunk * kChunk + (within - bx * kChunk);
  if (chunk * kChunk + (within - bx * kChunk) >= y_per || by >= y_all) return;
  const int m0 = by * kBM, n0 = bx * kBN;
  const int expert = __builtin_amdgcn_readfirstlane(expert_ids[by]);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;           // this wave's 64 x 64 sub-tile: rows wm*64, columns wn*64
  const int col = lane & 31, hi = lane >> 5;

  // ---- DMA duty: pieces p = wave*6 .. wave*6+5 of the 48 of a stage; piece p covers LDS bytes [p KiB, p KiB + 1 KiB) =
  // 8 rows of 128 bytes; pieces 0-15 are A rows, 16-47 W rows.  Lane l: row p*8 + l/8, position l%8 holds chunk
  // (l%8) ^ ((row >> 1) & 7) of that row's 64 k.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)((int64_t)a_rows * K * 2), 0x00020000);
  const T* w_tile = w + ((int64_t)expert * N + n0) * K;
  const int n_rows = min(kBN, N - n0);               // weight rows of this tile that exist
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_tile, 0, (int)((int64_t)n_rows * K * 2), 0x00020000);
  int voff[kPieces];
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int p = wave * kPieces + j;
    const int row = (p & 15) * 8 + (p < 16 ? 0 : (p >> 4) * 128 - 128) + (lane >> 3);   // row inside A (0..127) or W (0..255)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (p < 16) {
      const int sid = sorted_ids[m0 + row];
      const int tok = sid < num_valid ? sid / top_k_div : 0;      // padding rows read row 0 (finite; never stored)
      voff[j] = tok * K * 2 + chunk * 16;
    } else {
      voff[j] = min(row, n_rows - 1) * K * 2 + chunk * 16;        // columns past N read the last row (never stored)
    }
  }
  auto issue = [=](int kb) __attribute__((always_inline)) {
    const int stage = kb % kRing;
    const int koff = __builtin_amdgcn_readfirstlane(kb * kBK * 2);
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
      const int p = wave * kPieces + j;
      const uint32_t dst = lds0 + stage * kStage + p * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(p < 16 ? rsrc_a : rsrc_w, (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16,
                                               voff[j], koff, 0, 0);
    }
  };

  // ---- fragment addresses: A operand rows wm*64 + mt*32 + col, W operand rows wn*64 + nt*32 + col; chunk ks*2 + hi
  uint32_t a_addr[4], b_addr[4];   // per k-step (the position XOR depends on it); mt / nt / stage offsets are immediates or adds
  {
    const int ra = wm * 64 + col, rb = wn * 64 + col;   // +32 rows keeps (row >> 1) & 7 (32 is a multiple of 16)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      a_addr[ks] = lds0 + ra * kRowB + (((ks * 2 + hi) ^ ((ra >> 1) & 7)) * 16);
      b_addr[ks] = lds0 + kAStage + rb * kRowB + (((ks * 2 + hi) ^ ((rb >> 1) & 7)) * 16);
    }
  }

  // S^T form as in the attention kernels: the WEIGHT fragment is the A operand, the token fragment the B operand, so that
  // a lane holds ONE token row (m = lane & 31) and 16 output columns in runs of four: 8-byte stores, one sorted id and
  // one routed weight per lane and 32-row block instead of sixteen.
  f32x16 acc[2][2];   // [nt][mt]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkb = K / kBK;
  issue(0);
  if (nkb > 1) issue(1);
  for (int kb = 0; kb < nkb; ++kb) {
    // block kb has landed for this wave when at most the pieces of block kb + 1 are outstanding
    if (kb + 1 < nkb) wait_vm<kPieces>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kb + 2 < nkb) issue(kb + 2);     // into the stage block kb - 1 was read from (everyone is past this barrier)
    const uint32_t so = (uint32_t)((kb % kRing) * kStage);
    Frag16 af[2][2], bf[2][2];           // [buffer][mt / nt]: the next k-step's fragments fly while this one multiplies
    af[0][0].u = lds_read16<0>(a_addr[0] + so);
    af[0][1].u = lds_read16<32 * kRowB>(a_addr[0] + so);
    bf[0][0].u = lds_read16<0>(b_addr[0] + so);
    bf[0][1].u = lds_read16<32 * kRowB>(b_addr[0] + so);
    static_for<0, 4>([&](auto ks_c) {
      constexpr int KS = decltype(ks_c)::value, CUR = KS & 1, NXT = CUR ^ 1;
      if constexpr (KS < 3) {
        af[NXT][0].u = lds_read16<0>(a_addr[KS + 1] + so);
        af[NXT][1].u = lds_read16<32 * kRowB>(a_addr[KS + 1] + so);
        bf[NXT][0].u = lds_read16<0>(b_addr[KS + 1] + so);
        bf[NXT][1].u = lds_read16<32 * kRowB>(b_addr[KS + 1] + so);
        wait_lgkm<4>();
      } else {
        wait_lgkm<0>();
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = Mfma<T>::mma(as_frag<T>(bf[CUR][nt]), as_frag<T>(af[CUR][mt]), acc[nt][mt]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- epilogue: lane holds token row m = m0 + wm*64 + mt*32 + col and columns n0 + wn*64 + nt*32 + 8*q + 4*hi + (0..3)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int sid = sorted_ids[m0 + wm * 64 + mt * 32 + col];
    if (sid >= num_valid) continue;
    const float rw = mul_routed_weight ? topk_weights[sid] : 1.f;
    T* crow = c + (int64_t)sid * N + n0 + wn * 64 + 4 * hi;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = nt * 32 + 8 * q;
        if (n0 + wn * 64 + 4 * hi + nn < N) {
          uint2 pk;
          pk.x = pack2<T>(acc[nt][mt][q * 4 + 0] * rw, acc[nt][mt][q * 4 + 1] * rw);
          pk.y = pack2<T>(acc[nt][mt][q * 4 + 2] * rw, acc[nt][mt][q * 4 + 3] * rw);
          *reinterpret_cast<uint2*>(crow + nn) = pk;
        }
      }
    }
  }
}

// 0 = launched, 1 = shape not covered (the caller keeps the streaming kernel)
template <typename T>
int launch_moe_tiled_gemm(T* c, const T* a, const T* w, const float* topk_weights, const int32_t* sorted_ids,
                          const int32_t* expert_ids, const int32_t* num_post_pad, int64_t num_valid, int64_t n, int64_t k,
                          int64_t max_sorted, int top_k_div, int mul_routed_weight, hipStream_t st) {
  using namespace mtg;
  const int64_t a_rows = (num_valid + top_k_div - 1) / top_k_div;
  if (k % kBK != 0 || n % 32 != 0) return 1;
  if (a_rows * k * 2 >= (1ll << 31) || (int64_t)kBN * k * 2 >= (1ll << 31) || num_valid >= (1ll << 31)) return 1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)moe_tiled_gemm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    attr_set = true;
  }
  // (a block that starts below *num_tokens_post_pad, a multiple of 128 that is at most max_sorted, ends inside sorted_ids)
  const int64_t gx = (n + kBN - 1) / kBN, y_max = (max_sorted + kBM - 1) / kBM;
  const int64_t y_per = (y_max + 7) / 8, chunks = (y_per + kChunk - 1) / kChunk;
  dim3 grid((unsigned)(8 * chunks * kChunk * gx));
  hipLaunchKernelGGL((moe_tiled_gemm_kernel<T>), grid, dim3(512), kLds, st, c, a, w, topk_weights, sorted_ids, expert_ids,
                     num_post_pad, (int)num_valid, (int)n, (int)k, top_k_div, mul_routed_weight, (int)a_rows);
  return 0;
}

template int launch_moe_tiled_gemm<bf16_t>(bf16_t*, const bf16_t*, const bf16_t*, const float*, const int32_t*, const int32_t*,
                                           const int32_t*, int64_t, int64_t, int64_t, int64_t, int, int, hipStream_t);
template int launch_moe_tiled_gemm<f16_t>(f16_t*, const f16_t*, const f16_t*, const float*, const int32_t*, const int32_t*,
                                          const int32_t*, int64_t, int64_t, int64_t, int64_t, int, int, hipStream_t);

}  // namespace semipd
