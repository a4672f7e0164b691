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

#include <cstdlib>
#include <type_traits>

namespace semipd {

namespace mtg {

constexpr int kBM = 128;
constexpr int kChunk = 4;                            // row blocks per L2-resident chunk (see the tile mapping)

// NT = 32-column MFMA tiles per wave (2: 128 x 256 workgroup tile, 4: 128 x 512), BK = k per ring stage, RING stages
template <int NT, int BK, int RING>
struct Cfg {
  static constexpr int kBN = 4 * NT * 32;                     // 4 waves along n
  static constexpr int kRowB = BK * 2;                        // bytes of one LDS row
  static constexpr int kCPR = BK / 8;                         // 16-byte chunks per row
  static constexpr int kRPL = 256 / kRowB;                    // rows per 256-byte bank line
  static constexpr int kRPP = 1024 / kRowB;                   // rows per 1-KiB DMA piece
  static constexpr int kAStage = kBM * kRowB, kBStage = kBN * kRowB, kStage = kAStage + kBStage;
  static constexpr int kLds = RING * kStage;
  static constexpr int kAPieces = kAStage / 1024, kPiecesAll = kStage / 1024;
  static constexpr int kPieces = kPiecesAll / 8;              // per wave and k-block
  static constexpr int kKS = BK / 16;                         // MFMA k-steps per stage
  static_assert(kPiecesAll % 8 == 0 && kLds <= 160 * 1024, "whole pieces per wave; the ring must fit the LDS");
};

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

#define MTG_KERNEL_NAME moe_tiled_gemm_kernel_128x256
#define MTG_NT 2
#define MTG_BK 64
#define MTG_RING 3
#define MTG_SILU 0
#include "moe_tiled_gemm_kernel.inc"
#undef MTG_KERNEL_NAME
#undef MTG_SILU
#define MTG_KERNEL_NAME moe_tiled_gemm_silu_kernel_128x128
#define MTG_SILU 1
#include "moe_tiled_gemm_kernel.inc"
#undef MTG_KERNEL_NAME
#undef MTG_NT
#undef MTG_BK
#undef MTG_RING
#undef MTG_SILU
#define MTG_KERNEL_NAME moe_tiled_gemm_kernel_128x512
#define MTG_NT 4
#define MTG_BK 32
#define MTG_RING 4
#define MTG_SILU 0
#include "moe_tiled_gemm_kernel.inc"
#undef MTG_KERNEL_NAME
#undef MTG_NT
#undef MTG_BK
#undef MTG_RING
#undef MTG_SILU

// 0 = launched, 1 = shape not covered (the caller keeps the streaming kernel)
template <typename T>
int launch_moe_tiled_gemm(T* c, const T* a, const T* w, const float* topk_weights, const int32_t* sorted_ids,
                          const int32_t* expert_ids, const int32_t* num_post_pad, int64_t num_valid, int64_t n, int64_t k,
                          int64_t max_sorted, int top_k_div, int mul_routed_weight, hipStream_t st) {
  using namespace mtg;
  const int64_t a_rows = (num_valid + top_k_div - 1) / top_k_div;
  if (k % 64 != 0 || n % 32 != 0) return 1;
  if (a_rows * k * 2 >= (1ll << 31) || (int64_t)512 * k * 2 >= (1ll << 31) || num_valid >= (1ll << 31)) return 1;
  static const int form = [] { const char* e = getenv("SEMIPD_MOE_TILED_FORM"); return e ? atoi(e) : 0; }();   // 1: 128 x 256, 2: 128 x 512
  // (a block that starts below *num_tokens_post_pad, a multiple of 128 that is at most max_sorted, ends inside sorted_ids)
  const int64_t y_max = (max_sorted + kBM - 1) / kBM;
  const int64_t y_per = (y_max + 7) / 8, chunks = (y_per + kChunk - 1) / kChunk;
#define MTG(KERNEL, NTV, BKV, RV)                                                                                      \
  do {                                                                                                                 \
    using CF = Cfg<NTV, BKV, RV>;                                                                                      \
    static std::atomic<uint64_t> lds_ok{0};                                                                             \
    if (ensure_dynamic_lds((const void*)KERNEL<T>, CF::kLds, lds_ok, "moe_tiled_gemm")) return 1;                        \
    const int64_t gx = (n + CF::kBN - 1) / CF::kBN;                                                                    \
    dim3 grid((unsigned)(8 * chunks * kChunk * gx));                                                                   \
    hipLaunchKernelGGL((KERNEL<T>), grid, dim3(512), CF::kLds, st, c, a, w, topk_weights,  \
                       sorted_ids, expert_ids, num_post_pad, (int)num_valid, (int)n, (int)k, top_k_div,                 \
                       mul_routed_weight, (int)a_rows);                                                                 \
  } while (0)
  // 128 x 512 (0.75 KiB of fragments per MFMA instead of 1, stages of 32 k) measured 5-12 % slower than 128 x 256 on the
  // DeepSeek-V2-Lite shapes (profiles/r02_kbench_moe_tiled_gemm.txt): kept selectable, not the default
  if (form == 2) MTG(moe_tiled_gemm_kernel_128x512, 4, 32, 4);
  else MTG(moe_tiled_gemm_kernel_128x256, 2, 64, 3);
#undef MTG
  return 0;
}

// GEMM1 of the fused MoE with SiluAndMul in the epilogue: c [num_valid, n / 2] = silu(T(A W_gate^T)) * T(A W_up^T), w = [E, n, k]
// with the gate rows first.  0 = launched, 1 = shape not covered.
bool moe_tiled_gemm_silu_supported(int64_t num_valid, int64_t n, int64_t k, int top_k_div) {
  const int64_t a_rows = (num_valid + top_k_div - 1) / top_k_div;
  return k % 64 == 0 && n % 64 == 0 && a_rows * k * 2 < (1ll << 31) && (n / 2 + 128) * k * 2 < (1ll << 31) &&
         num_valid < (1ll << 31);
}

template <typename T>
int launch_moe_tiled_gemm_silu(T* c, const T* a, const T* w, const int32_t* sorted_ids, const int32_t* expert_ids,
                               const int32_t* num_post_pad, int64_t num_valid, int64_t n, int64_t k, int64_t max_sorted,
                               int top_k_div, hipStream_t st) {
  using namespace mtg;
  if (!moe_tiled_gemm_silu_supported(num_valid, n, k, top_k_div)) return 1;
  using CF = Cfg<2, 64, 3>;
  static std::atomic<uint64_t> lds_ok{0};
  if (ensure_dynamic_lds((const void*)moe_tiled_gemm_silu_kernel_128x128<T>, CF::kLds, lds_ok, "moe_tiled_gemm_silu")) return 1;
  const int64_t a_rows = (num_valid + top_k_div - 1) / top_k_div;
  const int64_t y_max = (max_sorted + kBM - 1) / kBM, y_per = (y_max + 7) / 8, chunks = (y_per + kChunk - 1) / kChunk;
  const int64_t gx = (n / 2 + CF::kBN / 2 - 1) / (CF::kBN / 2);
  dim3 grid((unsigned)(8 * chunks * kChunk * gx));
  hipLaunchKernelGGL((moe_tiled_gemm_silu_kernel_128x128<T>), grid, dim3(512), CF::kLds, st, c, a, w, (const float*)nullptr,
                     sorted_ids, expert_ids, num_post_pad, (int)num_valid, (int)n, (int)k, top_k_div, 0, (int)a_rows);
  return 0;
}

template int launch_moe_tiled_gemm_silu<bf16_t>(bf16_t*, const bf16_t*, const bf16_t*, const int32_t*, const int32_t*,
                                                const int32_t*, int64_t, int64_t, int64_t, int64_t, int, hipStream_t);
template int launch_moe_tiled_gemm_silu<f16_t>(f16_t*, const f16_t*, const f16_t*, const int32_t*, const int32_t*,
                                               const int32_t*, int64_t, int64_t, int64_t, int64_t, int, hipStream_t);

template int launch_moe_tiled_gemm<bf16_t>(bf16_t*, const bf16_t*, const bf16_t*, const float*, const int32_t*, const int32_t*,
                                           const int32_t*, int64_t, int64_t, int64_t, int64_t, int, int, hipStream_t);
template int launch_moe_tiled_gemm<f16_t>(f16_t*, const f16_t*, const f16_t*, const float*, const int32_t*, const int32_t*,
                                          const int32_t*, int64_t, int64_t, int64_t, int64_t, int, int, hipStream_t);

}  // namespace semipd
