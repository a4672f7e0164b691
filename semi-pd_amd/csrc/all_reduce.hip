// Tensor-parallel SUM all-reduce over peer-mapped memory: every rank reads the other
// ranks' payload directly through xGMI (or, for ranks that share a device, through the same HBM) and
// sums in fp32 in rank order, so all ranks produce bit-identical results.
//
// Follows the behaviour of sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh (one-stage kernel
// :260-284, two-stage reduce-scatter + all-gather :295-343, fp32 accumulation :123-153, size
// dispatch :538-552) with a different buffer protocol, chosen for gfx950:
//  * no buffer registration.  The kernel first copies its input into the rank's own slot of a shared,
//    UNCACHED region (hipDeviceMallocUncached, what allocate_meta_buffer does for the signals,
//    custom_all_reduce.hip:156-172) and peers read it from there.  The local copy costs HBM
//    bandwidth (8 TB/s) next to a transfer bound by xGMI links (153 GB/s each), and it removes the
//    per-graph IPC handle exchange (get_graph_buffer_ipc_meta / register_graph_buffers): any input
//    pointer works under hipGraph capture.
//  * the slots are double buffered on the call number (kept in device memory, so hipGraph replays
//    advance it) and every flag carries that number, so there is no closing barrier: a rank can be at
//    most one call ahead of its peers (it needs their flags of call n to leave call n, and their
//    kernel n + 1 cannot start before their kernel n is done), and call n + 1 writes the other buffer.
//  * block b of every rank talks only to block b of the peers (flags per block), so nothing ever
//    waits for a whole grid.
//  * xGMI is a full mesh of point-to-point links: a thread issues the loads to all peers back to back
//    before it uses any of them, so the 7 links of a GPU carry traffic at the same time; the sum always
//    runs over ranks 0..n-1.
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace semipd {

constexpr int kArMaxRanks = 8;
constexpr int kArMaxBlocks = 64;
constexpr int kArThreads = 512;
constexpr size_t kArLine = 256;  // chunk granularity in bytes: no cache line is shared by two blocks

struct ArSignal {
  uint32_t call;                              // all-reduce calls this rank has completed (own use)
  uint32_t done;                              // blocks of the running call that have finished (own use)
  uint32_t timed_out;                         // waits that gave up (only when a timeout is set; own use)
  uint32_t pad[61];
  uint32_t start[kArMaxBlocks][kArMaxRanks];  // [block][peer]: peer has staged its payload of call n
  uint32_t mid[kArMaxBlocks][kArMaxRanks];    // [block][peer]: peer has reduced its slice of call n
};
constexpr size_t kArMetaBytes = (sizeof(ArSignal) + 4095) / 4096 * 4096;

struct ArPeers {
  char* region[kArMaxRanks];  // peer-mapped base of every rank's shared region (own region at [rank])
  size_t max_bytes;           // payload capacity of one slot
  uint64_t timeout_ticks;     // 100 MHz ticks a flag wait may take; 0 = wait for ever (normal operation)
  uint32_t* cu_trace;         // tests: [8 XCCs][256 hardware CU ids] -- every block of every kernel marks where it ran; null = off
  int rank;
};

struct ArComm {
  ArPeers peers;
  int world;
  int device;
  int max_blocks;
  size_t one_shot_below;  // bytes: smaller payloads use the one-stage kernel
};

// region layout: [ArSignal | in slot 0 | in slot 1 | reduced slot 0 | reduced slot 1]
__device__ __forceinline__ ArSignal* ar_signal(const ArPeers& p, int r) { return reinterpret_cast<ArSignal*>(p.region[r]); }
__device__ __forceinline__ uint4* ar_in_slot(const ArPeers& p, int r, uint32_t buf) {
  return reinterpret_cast<uint4*>(p.region[r] + kArMetaBytes + (size_t)buf * p.max_bytes);
}
__device__ __forceinline__ uint4* ar_red_slot(const ArPeers& p, int r, uint32_t buf) {
  return reinterpret_cast<uint4*>(p.region[r] + kArMetaBytes + (size_t)(2 + buf) * p.max_bytes);
}

// All stores of the block become visible to the peers, then peers are told, then their flags are
// awaited, then nothing of theirs read before this point may be reused.
template <int NR>
__device__ __forceinline__ void ar_block_barrier(const ArPeers& p, uint32_t (ArSignal::*flags)[kArMaxBlocks][kArMaxRanks], uint32_t seq) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (threadIdx.x < NR) {
    const int peer = threadIdx.x;
    uint32_t* theirs = &(ar_signal(p, peer)->*flags)[blockIdx.x][p.rank];
    __hip_atomic_store(theirs, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = &(ar_signal(p, p.rank)->*flags)[blockIdx.x][peer];
    const uint64_t t0 = p.timeout_ticks ? wall_clock64() : 0;
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (p.timeout_ticks && wall_clock64() - t0 > p.timeout_ticks) {
        // self-test mode: report instead of hanging; the payload of this call is then undefined
        __hip_atomic_fetch_add(&ar_signal(p, p.rank)->timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

template <typename T, int NR>
__device__ __forceinline__ uint4 ar_sum(const uint4 (&v)[NR]) {
  constexpr int E = Elem<T>::kVec;
  float acc[E];
  const T* e0 = reinterpret_cast<const T*>(&v[0]);
#pragma unroll
  for (int i = 0; i < E; ++i) acc[i] = Elem<T>::to_f(e0[i]);
#pragma unroll
  for (int r = 1; r < NR; ++r) {
    const T* e = reinterpret_cast<const T*>(&v[r]);
#pragma unroll
    for (int i = 0; i < E; ++i) acc[i] += Elem<T>::to_f(e[i]);
  }
  uint4 out;
  T* o = reinterpret_cast<T*>(&out);
#pragma unroll
  for (int i = 0; i < E; ++i) o[i] = Elem<T>::from_f(acc[i]);
  return out;
}

// The call number is the same for every block of a launch: it only moves when the last block of the
// previous launch has finished, and launches of one rank are ordered by its stream.
__device__ __forceinline__ uint32_t ar_begin(const ArPeers& p) {
  __shared__ uint32_t s_seq;
  if (threadIdx.x == 0 && p.cu_trace) {
    // HW_REG_XCC_ID (id 20) bits [3:0]; HW_REG_HW_ID (id 4): CU id bits [11:8], SH bit 12, SE bits [15:13] (csrc/ipc.hip's probe)
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf;
    const unsigned hw = (__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) >> 8) & 0xff;
    p.cu_trace[(xcc & 7) * 256 + hw] = 1u;
  }
  if (threadIdx.x == 0) s_seq = __hip_atomic_load(&ar_signal(p, p.rank)->call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  __syncthreads();
  return s_seq;
}

__device__ __forceinline__ void ar_end(const ArPeers& p, uint32_t seq) {
  if (threadIdx.x == 0) {
    ArSignal* me = ar_signal(p, p.rank);
    const uint32_t finished = __hip_atomic_fetch_add(&me->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (finished == gridDim.x - 1) {
      __hip_atomic_store(&me->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&me->call, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One stage: stage, barrier, every rank sums the whole payload.  nvec = 16-byte vectors, chunk = the
// vectors one block owns (a multiple of kArLine / 16).
template <typename T, int NR>
__global__ void __launch_bounds__(kArThreads) ar_one_stage_kernel(ArPeers p, const uint4* __restrict__ in,
                                                                  uint4* __restrict__ out, size_t nvec, size_t chunk) {
  const uint32_t seq = ar_begin(p);
  const uint32_t buf = seq & 1u;
  const size_t lo = (size_t)blockIdx.x * chunk;
  const size_t hi = lo + chunk < nvec ? lo + chunk : nvec;
  uint4* mine = ar_in_slot(p, p.rank, buf);
  for (size_t i = lo + threadIdx.x; i < hi; i += kArThreads) mine[i] = in[i];
  ar_block_barrier<NR>(p, &ArSignal::start, seq);
  ar_end(p, seq);
  const uint4* src[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) src[r] = ar_in_slot(p, r, buf);
  for (size_t i = lo + threadIdx.x; i < hi; i += kArThreads) {
    uint4 v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = src[r][i];
    out[i] = ar_sum<T, NR>(v);
  }
}

// Two stages: rank r reduces slice r (part vectors) into its reduced slot, peers gather it.  Block b
// owns sub-chunk b (sub vectors) of EVERY slice, in all three phases.
template <typename T, int NR>
__global__ void __launch_bounds__(kArThreads) ar_two_stage_kernel(ArPeers p, const uint4* __restrict__ in,
                                                                  uint4* __restrict__ out, size_t nvec, size_t part,
                                                                  size_t sub) {
  const uint32_t seq = ar_begin(p);
  const uint32_t buf = seq & 1u;
  const size_t s_lo = (size_t)blockIdx.x * sub;
  const size_t s_hi = s_lo + sub < part ? s_lo + sub : part;
  uint4* mine = ar_in_slot(p, p.rank, buf);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const size_t base = (size_t)r * part;
    for (size_t i = s_lo + threadIdx.x; i < s_hi && base + i < nvec; i += kArThreads) mine[base + i] = in[base + i];
  }
  ar_block_barrier<NR>(p, &ArSignal::start, seq);
  {
    const size_t base = (size_t)p.rank * part;
    const uint4* src[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) src[r] = ar_in_slot(p, r, buf) + base;
    uint4* red = ar_red_slot(p, p.rank, buf);
    for (size_t i = s_lo + threadIdx.x; i < s_hi && base + i < nvec; i += kArThreads) {
      uint4 v[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) v[r] = src[r][i];
      red[i] = ar_sum<T, NR>(v);
    }
  }
  ar_block_barrier<NR>(p, &ArSignal::mid, seq);
  ar_end(p, seq);
  for (int k = 1; k <= NR; ++k) {  // start with the next rank: spreads the gather over the links
    const int r = (p.rank + k) % NR;
    const size_t base = (size_t)r * part;
    const uint4* red = ar_red_slot(p, r, buf);
    for (size_t i = s_lo + threadIdx.x; i < s_hi && base + i < nvec; i += kArThreads) out[base + i] = red[i];
  }
}

// All-gather with the same staging protocol: out[r * nvec + i] = rank r's in[i].
template <int NR>
__global__ void __launch_bounds__(kArThreads) ar_all_gather_kernel(ArPeers p, const uint4* __restrict__ in,
                                                                   uint4* __restrict__ out, size_t nvec, size_t chunk) {
  const uint32_t seq = ar_begin(p);
  const uint32_t buf = seq & 1u;
  const size_t lo = (size_t)blockIdx.x * chunk;
  const size_t hi = lo + chunk < nvec ? lo + chunk : nvec;
  uint4* mine = ar_in_slot(p, p.rank, buf);
  for (size_t i = lo + threadIdx.x; i < hi; i += kArThreads) mine[i] = in[i];
  ar_block_barrier<NR>(p, &ArSignal::start, seq);
  ar_end(p, seq);
  for (int k = 1; k <= NR; ++k) {
    const int r = (p.rank + k) % NR;
    const uint4* src = ar_in_slot(p, r, buf);
    uint4* dst = out + (size_t)r * nvec;
    for (size_t i = lo + threadIdx.x; i < hi; i += kArThreads) dst[i] = src[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Expert-parallel all-to-all (BASELINE config 5) over the same peer-mapped regions and call sequence.
// The reference has none (its expert parallelism keeps every token on every rank and all-reduces the partial outputs,
// layers/moe/ep_moe/layer.py:190): this is new design, defined by oracle/ops.py: ep_dispatch / ep_combine.
//   dispatch: rank s stages its token rows, their routed expert ids and routing weights in its slot; rank d PULLS one row
//             per (token, j) entry routed to one of its experts, ordered by sender rank, token, j -- index and byte work;
//   combine:  rank d stages its expert-output rows; rank s pulls, for each of its tokens, the k rows that belong to it and
//             sums them in j order (fp32, one rounding) -- the arithmetic of moe_sum (csrc/elementwise.hip).
// Pull, not push: a reader issues loads to all peers back to back over the 7 xGMI links, like the all-reduce above, and
// nobody writes into memory another rank may still be reading.  Staging is a separate local launch in front of the
// exchange kernel; the exchange kernel's per-block flag exchange (ar_block_barrier) then proves the peer's staging launch
// complete (stream order on the peer), and the double buffering on the call number covers the way out as it does above.
struct EpHeader {          // first 256 bytes of a staged slot
  int32_t tokens, top_k, rows, pad;
  int32_t counts[kArMaxRanks];   // entries this rank sends to each destination (dispatch) / unused (combine)
  int32_t pad2[64 - 4 - kArMaxRanks];
};
static_assert(sizeof(EpHeader) == 256, "EpHeader is one 256-byte line");

__device__ __forceinline__ size_t ep_align(size_t x) { return (x + 255) / 256 * 256; }
// dispatch slot layout: [EpHeader | within int32[T k] | ids int32[T k] | w float[T k] | rows T x row_bytes], 256-aligned parts
__device__ __forceinline__ char* ep_part(char* slot, int part, size_t entries) {
  return slot + 256 + (size_t)part * ep_align(entries * 4);
}

// One workgroup, local: destination and position-within-(sender, destination) of every (token, j) entry in (t, j) order,
// the per-destination counts, all staged in this rank's slot of the NEXT call together with the ids and weights.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) ep_prepare_kernel(ArPeers p, const int32_t* __restrict__ ids,
                                                            const float* __restrict__ w, int tokens, int top_k,
                                                            int experts_per_rank, int world, int32_t* __restrict__ within_out) {
  __shared__ int32_t cnt[THREADS][kArMaxRanks];
  __shared__ int32_t total[kArMaxRanks];
  const uint32_t buf = (__hip_atomic_load(&ar_signal(p, p.rank)->call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1) & 1u;
  char* slot = reinterpret_cast<char*>(ar_in_slot(p, p.rank, buf));
  const int entries = tokens * top_k;
  const int per = (entries + THREADS - 1) / THREADS;
  const int lo = min(entries, (int)threadIdx.x * per), hi = min(entries, lo + per);
  int32_t mine[kArMaxRanks];
#pragma unroll
  for (int d = 0; d < kArMaxRanks; ++d) mine[d] = 0;
  for (int e = lo; e < hi; ++e) {
    const int d = ids[e] / experts_per_rank;
#pragma unroll
    for (int q = 0; q < kArMaxRanks; ++q) mine[q] += (q == d);
  }
#pragma unroll
  for (int d = 0; d < kArMaxRanks; ++d) cnt[threadIdx.x][d] = mine[d];
  __syncthreads();
  if (threadIdx.x < kArMaxRanks) {   // exclusive scan over the threads' chunks, one destination per thread
    int32_t run = 0;
    for (int t = 0; t < THREADS; ++t) {
      const int32_t c = cnt[t][threadIdx.x];
      cnt[t][threadIdx.x] = run;
      run += c;
    }
    total[threadIdx.x] = run;
  }
  __syncthreads();
  int32_t* s_within = reinterpret_cast<int32_t*>(ep_part(slot, 0, entries));
  int32_t* s_ids = reinterpret_cast<int32_t*>(ep_part(slot, 1, entries));
  float* s_w = reinterpret_cast<float*>(ep_part(slot, 2, entries));
  int32_t base[kArMaxRanks];
#pragma unroll
  for (int d = 0; d < kArMaxRanks; ++d) base[d] = cnt[threadIdx.x][d];
  for (int e = lo; e < hi; ++e) {
    const int id = ids[e];
    const int d = id / experts_per_rank;
    int32_t pos = 0;
#pragma unroll
    for (int q = 0; q < kArMaxRanks; ++q)
      if (q == d) pos = base[q]++;
    s_within[e] = pos;
    within_out[e] = pos;
    s_ids[e] = id;
    s_w[e] = w[e];
  }
  if (threadIdx.x == 0) {
    EpHeader* h = reinterpret_cast<EpHeader*>(slot);
    h->tokens = tokens;
    h->top_k = top_k;
    h->rows = tokens;
    for (int d = 0; d < kArMaxRanks; ++d) h->counts[d] = d < world ? total[d] : 0;
  }
}

// local: rows [n_rows, row_bytes] -> this rank's slot of the NEXT call at byte offset `off`; n_rows read from the device
// when `n_rows_dev` is given (the rows an earlier dispatch delivered).  With hdr != 0 also the header of a combine slot.
__global__ void __launch_bounds__(256) ep_stage_rows_kernel(ArPeers p, const uint4* __restrict__ rows, int64_t n_rows,
                                                           const int32_t* __restrict__ n_rows_dev, int64_t max_rows,
                                                           int64_t row_vecs, size_t off, int hdr) {
  const uint32_t buf = (__hip_atomic_load(&ar_signal(p, p.rank)->call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1) & 1u;
  char* slot = reinterpret_cast<char*>(ar_in_slot(p, p.rank, buf));
  if (n_rows_dev) n_rows = min((int64_t)n_rows_dev[0], max_rows);
  if (hdr && blockIdx.x == 0 && threadIdx.x == 0) {
    EpHeader* h = reinterpret_cast<EpHeader*>(slot);
    h->tokens = 0;
    h->top_k = 0;
    h->rows = (int32_t)n_rows;
  }
  uint4* dst = reinterpret_cast<uint4*>(slot + off);
  const int64_t total = n_rows * row_vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = rows[i];
}

// exchange: rank `me` pulls the entries routed to its experts from every sender's staged slot
template <int NR>
__global__ void __launch_bounds__(kArThreads) ep_dispatch_pull_kernel(ArPeers p, int experts_per_rank, int64_t row_vecs,
                                                                      uint4* __restrict__ recv_x, int32_t* __restrict__ recv_expert,
                                                                      float* __restrict__ recv_w, int64_t max_recv,
                                                                      int32_t* __restrict__ recv_count,
                                                                      int32_t* __restrict__ counts_all) {
  const uint32_t seq = ar_begin(p);
  const uint32_t buf = seq & 1u;
  ar_block_barrier<NR>(p, &ArSignal::start, seq);
  ar_end(p, seq);
  const int me = p.rank;
  __shared__ int32_t s_off[NR], s_entries[NR], s_topk[NR];
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int s = 0; s < NR; ++s) {
      const EpHeader* h = reinterpret_cast<const EpHeader*>(ar_in_slot(p, s, buf));
      s_off[s] = run;
      run += h->counts[me];
      s_entries[s] = h->tokens * h->top_k;
      s_topk[s] = h->top_k;
      if (blockIdx.x == 0)
        for (int d = 0; d < NR; ++d) counts_all[s * NR + d] = h->counts[d];
    }
    if (blockIdx.x == 0) recv_count[0] = run;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * (kArThreads / 64) + (threadIdx.x >> 6), nw = gridDim.x * (kArThreads / 64);
  for (int k = 1; k <= NR; ++k) {            // start with the next rank: spreads the pulls over the links
    const int s = (me + k) % NR;
    char* slot = reinterpret_cast<char*>(ar_in_slot(p, s, buf));
    const int entries = s_entries[s], top_k = s_topk[s];
    const int32_t* within = reinterpret_cast<const int32_t*>(ep_part(slot, 0, entries));
    const int32_t* ids = reinterpret_cast<const int32_t*>(ep_part(slot, 1, entries));
    const float* w = reinterpret_cast<const float*>(ep_part(slot, 2, entries));
    const uint4* rows = reinterpret_cast<const uint4*>(ep_part(slot, 3, entries));
    for (int e = gw; e < entries; e += nw) {
      const int id = ids[e];
      if (id / experts_per_rank != me) continue;
      const int64_t pos = (int64_t)s_off[s] + within[e];
      if (pos >= max_recv) continue;          // the caller sees recv_count > max_recv
      const uint4* src = rows + (int64_t)(e / top_k) * row_vecs;
      uint4* dst = recv_x + pos * row_vecs;
      for (int64_t v = lane; v < row_vecs; v += 64) dst[v] = src[v];
      if (lane == 0) {
        recv_expert[pos] = id - me * experts_per_rank;
        recv_w[pos] = w[e];
      }
    }
  }
}

// the way back: out[t] = T(sum_j fp32(y_{dest(t, j)}[position of (t, j)])) in j order
template <typename T, int NR>
__global__ void __launch_bounds__(kArThreads) ep_combine_pull_kernel(ArPeers p, const int32_t* __restrict__ ids,
                                                                     const int32_t* __restrict__ within,
                                                                     const int32_t* __restrict__ counts_all, int tokens, int top_k,
                                                                     int experts_per_rank, int64_t row_vecs, int64_t max_recv,
                                                                     uint4* __restrict__ out) {
  constexpr int V = Elem<T>::kVec;
  const uint32_t seq = ar_begin(p);
  const uint32_t buf = seq & 1u;
  ar_block_barrier<NR>(p, &ArSignal::start, seq);
  ar_end(p, seq);
  const int me = p.rank;
  __shared__ int32_t s_off[NR];            // where this rank's entries start in destination d's list
  if (threadIdx.x < NR) {
    int32_t run = 0;
    for (int s = 0; s < me; ++s) run += counts_all[s * NR + threadIdx.x];
    s_off[threadIdx.x] = run;
  }
  __syncthreads();
  const uint4* ybase[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) ybase[r] = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(ar_in_slot(p, r, buf)) + 256);
  const int64_t total = (int64_t)tokens * row_vecs;
  for (int64_t i = (int64_t)blockIdx.x * kArThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kArThreads) {
    const int t = (int)(i / row_vecs);
    const int64_t c = i - (int64_t)t * row_vecs;
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    for (int j = 0; j < top_k; ++j) {
      const int e = t * top_k + j;
      const int d = ids[e] / experts_per_rank;
      const int64_t pos = (int64_t)s_off[d] + within[e];
      uint4 v = make_uint4(0, 0, 0, 0);
      // an entry the destination's dispatch dropped (position >= max_recv, the same bound on every rank) was never
      // computed and never staged: it contributes zero instead of whatever lies behind the staged rows
      if (pos < max_recv) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (r == d) v = ybase[r][pos * row_vecs + c];
      }
      const T* ev = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += Elem<T>::to_f(ev[q]);
    }
    uint4 o;
    T* eo = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int q = 0; q < V; ++q) eo[q] = Elem<T>::from_f(acc[q]);
    out[i] = o;
  }
}

static size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }
static size_t ep_dispatch_bytes(int64_t tokens, int top_k, int64_t row_bytes) {
  return 256 + 3 * round_up((size_t)tokens * top_k * 4, 256) + (size_t)tokens * row_bytes;
}

template <int NR>
static int ag_launch(const ArComm* c, const void* in, void* out, size_t bytes, hipStream_t stream) {
  const size_t nvec = bytes / 16;
  const size_t line = kArLine / 16;
  size_t blocks = (nvec + kArThreads - 1) / kArThreads;
  if (blocks > (size_t)c->max_blocks) blocks = c->max_blocks;
  const size_t chunk = round_up((nvec + blocks - 1) / blocks, line);
  blocks = (nvec + chunk - 1) / chunk;
  hipLaunchKernelGGL((ar_all_gather_kernel<NR>), dim3((unsigned)blocks), dim3(kArThreads), 0, stream, c->peers,
                     (const uint4*)in, (uint4*)out, nvec, chunk);
  return launch_status("ar_all_gather_kernel");
}

template <typename T, int NR>
static int ar_launch(const ArComm* c, const void* in, void* out, size_t bytes, hipStream_t stream) {
  const size_t nvec = bytes / 16;
  const size_t line = kArLine / 16;
  const bool one = NR == 2 || bytes < c->one_shot_below;
  if (one) {
    size_t blocks = (nvec + kArThreads - 1) / kArThreads;
    if (blocks > (size_t)c->max_blocks) blocks = c->max_blocks;
    if (blocks < 1) blocks = 1;
    const size_t chunk = round_up((nvec + blocks - 1) / blocks, line);
    blocks = (nvec + chunk - 1) / chunk;
    hipLaunchKernelGGL((ar_one_stage_kernel<T, NR>), dim3((unsigned)blocks), dim3(kArThreads), 0, stream, c->peers,
                       (const uint4*)in, (uint4*)out, nvec, chunk);
    return launch_status("ar_one_stage_kernel");
  }
  const size_t part = round_up((nvec + NR - 1) / NR, line);
  size_t blocks = (part + kArThreads - 1) / kArThreads;
  if (blocks > (size_t)c->max_blocks) blocks = c->max_blocks;
  const size_t sub = round_up((part + blocks - 1) / blocks, line);
  blocks = (part + sub - 1) / sub;
  hipLaunchKernelGGL((ar_two_stage_kernel<T, NR>), dim3((unsigned)blocks), dim3(kArThreads), 0, stream, c->peers,
                     (const uint4*)in, (uint4*)out, nvec, part, sub);
  return launch_status("ar_two_stage_kernel");
}

template <typename T>
static int ar_dispatch_world(const ArComm* c, const void* in, void* out, size_t bytes, hipStream_t stream) {
  switch (c->world) {
    case 2: return ar_launch<T, 2>(c, in, out, bytes, stream);
    case 4: return ar_launch<T, 4>(c, in, out, bytes, stream);
    case 6: return ar_launch<T, 6>(c, in, out, bytes, stream);
    case 8: return ar_launch<T, 8>(c, in, out, bytes, stream);
  }
  set_error("all_reduce: world size %d is not one of 2, 4, 6, 8", c->world);
  return SEMIPD_EINVAL;
}

static long env_long(const char* name, long dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atol(s) : dflt;
}

}  // namespace semipd

using namespace semipd;

extern "C" {

size_t semipd_ar_meta_size(void) { return kArMetaBytes; }

size_t semipd_ar_region_size(size_t max_bytes) { return kArMetaBytes + 4 * round_up(max_bytes, kArLine); }

int semipd_ar_alloc_shared(size_t bytes, void** ptr) {
  SEMIPD_CHECK_ARG(ptr && bytes > 0, SEMIPD_EINVAL, "ar_alloc_shared: null pointer or zero size");
  void* p = nullptr;
  SEMIPD_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
  hipError_t e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    set_error("ar_alloc_shared: clearing %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return (int)e;
  }
  *ptr = p;
  return 0;
}

int semipd_ar_free_shared(void* ptr) {
  if (ptr) SEMIPD_HIP(hipFree(ptr));
  return 0;
}

int semipd_ar_init(void* const* regions, size_t region_bytes, int rank, int world, void** comm) {
  SEMIPD_CHECK_ARG(regions && comm, SEMIPD_EINVAL, "ar_init: null pointer");
  SEMIPD_CHECK_ARG(world == 2 || world == 4 || world == 6 || world == 8, SEMIPD_EINVAL,
                   "ar_init: world size %d is not one of 2, 4, 6, 8", world);
  SEMIPD_CHECK_ARG(rank >= 0 && rank < world, SEMIPD_EINVAL, "ar_init: invalid rank %d of %d", rank, world);
  SEMIPD_CHECK_ARG(region_bytes > kArMetaBytes + 4 * kArLine, SEMIPD_EINVAL,
                   "ar_init: region of %zu bytes holds no payload (meta is %zu)", region_bytes, kArMetaBytes);
  ArComm* c = new ArComm();
  for (int r = 0; r < world; ++r) {
    if (!regions[r] || (reinterpret_cast<uintptr_t>(regions[r]) & 255u)) {
      delete c;
      set_error("ar_init: region of rank %d is null or not 256-byte aligned", r);
      return SEMIPD_EINVAL;
    }
    c->peers.region[r] = static_cast<char*>(regions[r]);
  }
  c->peers.max_bytes = (region_bytes - kArMetaBytes) / 4 / kArLine * kArLine;
  c->peers.rank = rank;
  c->peers.timeout_ticks = 0;
  c->peers.cu_trace = nullptr;
  c->world = world;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    delete c;
    set_error("ar_init: hipGetDevice failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  c->device = dev;
  long mb = env_long("SEMIPD_AR_MAX_BLOCKS", 32);
  c->max_blocks = (int)(mb < 1 ? 1 : (mb > kArMaxBlocks ? kArMaxBlocks : mb));
  // custom_all_reduce_hip.cuh:541-547: one stage below 512 KB for up to 4 ranks, below 256 KB for up to 8
  c->one_shot_below = (size_t)env_long("SEMIPD_AR_ONE_SHOT_BELOW", world <= 4 ? 512 * 1024 : 256 * 1024);
  *comm = c;
  return 0;
}

int semipd_ar_set_timeout_ms(void* comm, uint32_t ms) {
  SEMIPD_CHECK_ARG(comm, SEMIPD_EINVAL, "ar_set_timeout_ms: null pointer");
  static_cast<ArComm*>(comm)->peers.timeout_ticks = (uint64_t)ms * 100000ull;  // wall_clock64: 100 MHz
  return 0;
}

int semipd_ar_timed_out(void* comm, uint32_t* count) {
  SEMIPD_CHECK_ARG(comm && count, SEMIPD_EINVAL, "ar_timed_out: null pointer");
  const ArComm* c = static_cast<ArComm*>(comm);
  const ArSignal* me = reinterpret_cast<const ArSignal*>(c->peers.region[c->peers.rank]);
  SEMIPD_HIP(hipMemcpy(count, &me->timed_out, sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
}

int semipd_ar_max_bytes(void* comm, size_t* bytes) {
  SEMIPD_CHECK_ARG(comm && bytes, SEMIPD_EINVAL, "ar_max_bytes: null pointer");
  *bytes = static_cast<ArComm*>(comm)->peers.max_bytes;
  return 0;
}

int semipd_ar_all_reduce(void* comm, const void* in, void* out, size_t numel, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(comm && in && out, SEMIPD_EINVAL, "ar_all_reduce: null pointer");
  const ArComm* c = static_cast<ArComm*>(comm);
  const size_t esz = dtype == SEMIPD_F32 ? 4 : 2;
  SEMIPD_CHECK_ARG(dtype == SEMIPD_F32 || dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE,
                   "ar_all_reduce: dtype code %d is not f32 / bf16 / f16", dtype);
  const size_t bytes = numel * esz;
  // custom_all_reduce.py:451-453: the byte size must be a multiple of 16
  SEMIPD_CHECK_ARG(bytes > 0 && bytes % 16 == 0, SEMIPD_ESHAPE, "ar_all_reduce: %zu bytes is not a positive multiple of 16", bytes);
  SEMIPD_CHECK_ARG(bytes <= c->peers.max_bytes, SEMIPD_ESHAPE, "ar_all_reduce: %zu bytes exceed the registered capacity %zu",
                   bytes, c->peers.max_bytes);
  SEMIPD_CHECK_ARG(aligned16(in) && aligned16(out), SEMIPD_EALIGN, "ar_all_reduce: input and output must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case SEMIPD_F32: return ar_dispatch_world<float>(c, in, out, bytes, s);
    case SEMIPD_BF16: return ar_dispatch_world<bf16_t>(c, in, out, bytes, s);
    default: return ar_dispatch_world<f16_t>(c, in, out, bytes, s);
  }
}

int semipd_ar_all_gather(void* comm, const void* in, void* out, size_t bytes_per_rank, void* stream) {
  SEMIPD_CHECK_ARG(comm && in && out, SEMIPD_EINVAL, "ar_all_gather: null pointer");
  const ArComm* c = static_cast<ArComm*>(comm);
  SEMIPD_CHECK_ARG(bytes_per_rank > 0 && bytes_per_rank % 16 == 0, SEMIPD_ESHAPE,
                   "ar_all_gather: %zu bytes per rank is not a positive multiple of 16", bytes_per_rank);
  SEMIPD_CHECK_ARG(bytes_per_rank <= c->peers.max_bytes, SEMIPD_ESHAPE,
                   "ar_all_gather: %zu bytes per rank exceed the registered capacity %zu", bytes_per_rank, c->peers.max_bytes);
  SEMIPD_CHECK_ARG(aligned16(in) && aligned16(out), SEMIPD_EALIGN, "ar_all_gather: input and output must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (c->world) {
    case 2: return ag_launch<2>(c, in, out, bytes_per_rank, s);
    case 4: return ag_launch<4>(c, in, out, bytes_per_rank, s);
    case 6: return ag_launch<6>(c, in, out, bytes_per_rank, s);
    default: return ag_launch<8>(c, in, out, bytes_per_rank, s);
  }
}

int semipd_ep_dispatch(void* comm, const void* x, const int32_t* topk_ids, const float* topk_weights, int64_t tokens, int top_k,
                       int64_t row_bytes, int experts_per_rank, void* recv_x, int32_t* recv_expert, float* recv_weight,
                       int64_t max_recv, int32_t* recv_count, int32_t* send_within, int32_t* counts_all, void* stream) {
  SEMIPD_CHECK_ARG(comm && recv_x && recv_expert && recv_weight && recv_count && counts_all, SEMIPD_EINVAL,
                   "ep_dispatch: null pointer");
  SEMIPD_CHECK_ARG(tokens >= 0 && top_k > 0 && top_k <= 64 && experts_per_rank > 0 && max_recv >= 0, SEMIPD_EINVAL,
                   "ep_dispatch: bad sizes");
  SEMIPD_CHECK_ARG(tokens == 0 || (x && topk_ids && topk_weights && send_within), SEMIPD_EINVAL, "ep_dispatch: null input");
  SEMIPD_CHECK_ARG(row_bytes > 0 && row_bytes % 16 == 0 && aligned16(x) && aligned16(recv_x), SEMIPD_EALIGN,
                   "ep_dispatch: rows must be a multiple of 16 bytes and 16-byte aligned");
  const ArComm* c = static_cast<ArComm*>(comm);
  SEMIPD_CHECK_ARG(tokens * top_k < (1 << 30) && ep_dispatch_bytes(tokens, top_k, row_bytes) <= c->peers.max_bytes, SEMIPD_ESHAPE,
                   "ep_dispatch: %lld tokens of %lld bytes with top-%d need %zu bytes of staging, the region holds %zu",
                   (long long)tokens, (long long)row_bytes, top_k, ep_dispatch_bytes(tokens, top_k, row_bytes), c->peers.max_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int entries = (int)(tokens * top_k);
  hipLaunchKernelGGL((ep_prepare_kernel<512>), dim3(1), dim3(512), 0, s, c->peers, topk_ids, topk_weights, (int)tokens, top_k,
                     experts_per_rank, c->world, send_within);
  if (int rc = launch_status("ep_prepare_kernel")) return rc;
  if (tokens > 0) {
    const int64_t vecs = tokens * (row_bytes / 16);
    const unsigned blocks = (unsigned)std::min<int64_t>((vecs + 255) / 256, 1024);
    hipLaunchKernelGGL(ep_stage_rows_kernel, dim3(blocks), dim3(256), 0, s, c->peers, (const uint4*)x, tokens,
                       (const int32_t*)nullptr, tokens, row_bytes / 16, (size_t)256 + 3 * round_up((size_t)entries * 4, 256), 0);
    if (int rc = launch_status("ep_stage_rows_kernel")) return rc;
  }
#define EP_DISPATCH(NRV)                                                                                                   \
  hipLaunchKernelGGL((ep_dispatch_pull_kernel<NRV>), dim3((unsigned)c->max_blocks), dim3(kArThreads), 0, s, c->peers,      \
                     experts_per_rank, row_bytes / 16, (uint4*)recv_x, recv_expert, recv_weight, max_recv, recv_count, counts_all)
  switch (c->world) {
    case 2: EP_DISPATCH(2); break;
    case 4: EP_DISPATCH(4); break;
    case 6: EP_DISPATCH(6); break;
    default: EP_DISPATCH(8); break;
  }
#undef EP_DISPATCH
  return launch_status("ep_dispatch_pull_kernel");
}

int semipd_ep_combine(void* comm, const void* y, const int32_t* recv_count, int64_t max_recv, const int32_t* topk_ids,
                      const int32_t* send_within, const int32_t* counts_all, void* out, int64_t tokens, int top_k, int64_t hidden,
                      int experts_per_rank, int dtype, void* stream) {
  SEMIPD_CHECK_ARG(comm && y && recv_count && counts_all, SEMIPD_EINVAL, "ep_combine: null pointer");
  SEMIPD_CHECK_ARG(tokens >= 0 && top_k > 0 && hidden > 0 && experts_per_rank > 0 && max_recv >= 0, SEMIPD_EINVAL,
                   "ep_combine: bad sizes");
  SEMIPD_CHECK_ARG(tokens == 0 || (out && topk_ids && send_within), SEMIPD_EINVAL, "ep_combine: null pointer");
  SEMIPD_CHECK_ARG(dtype == SEMIPD_BF16 || dtype == SEMIPD_F16, SEMIPD_EDTYPE, "ep_combine: bf16 / f16 rows only");
  SEMIPD_CHECK_ARG(hidden % 8 == 0 && aligned16(y) && aligned16(out), SEMIPD_EALIGN,
                   "ep_combine: hidden must be a multiple of 8 and the rows 16-byte aligned");
  const ArComm* c = static_cast<ArComm*>(comm);
  const int64_t row_bytes = hidden * 2;
  SEMIPD_CHECK_ARG(256 + (size_t)max_recv * row_bytes <= c->peers.max_bytes, SEMIPD_ESHAPE,
                   "ep_combine: %lld rows of %lld bytes do not fit the region's %zu bytes", (long long)max_recv,
                   (long long)row_bytes, c->peers.max_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    const int64_t vecs = std::max<int64_t>(max_recv, 1) * (row_bytes / 16);
    const unsigned blocks = (unsigned)std::min<int64_t>((vecs + 255) / 256, 1024);
    hipLaunchKernelGGL(ep_stage_rows_kernel, dim3(blocks), dim3(256), 0, s, c->peers, (const uint4*)y, (int64_t)0, recv_count,
                       max_recv, row_bytes / 16, (size_t)256, 1);
    if (int rc = launch_status("ep_stage_rows_kernel")) return rc;
  }
#define EP_COMBINE(TT, NRV)                                                                                                 \
  hipLaunchKernelGGL((ep_combine_pull_kernel<TT, NRV>), dim3((unsigned)c->max_blocks), dim3(kArThreads), 0, s, c->peers,    \
                     topk_ids, send_within, counts_all, (int)tokens, top_k, experts_per_rank, row_bytes / 16, max_recv, (uint4*)out)
#define EP_COMBINE_W(TT)            \
  switch (c->world) {               \
    case 2: EP_COMBINE(TT, 2); break; \
    case 4: EP_COMBINE(TT, 4); break; \
    case 6: EP_COMBINE(TT, 6); break; \
    default: EP_COMBINE(TT, 8); break; \
  }
  if (dtype == SEMIPD_BF16) { EP_COMBINE_W(bf16_t) } else { EP_COMBINE_W(f16_t) }
#undef EP_COMBINE_W
#undef EP_COMBINE
  return launch_status("ep_combine_pull_kernel");
}

/* tests: every block of every kernel launched on this communicator from now on marks buf[xcc * 256 + hardware CU id] = 1
 * (buf: 2048 x uint32 of device memory, zeroed by the caller; NULL switches the trace off).  The evidence that collectives
 * launched on a CU-masked stream stay inside the mask (tests/test_gpu_comm_confined.py). */
int semipd_ar_set_cu_trace(void* comm, uint32_t* buf) {
  SEMIPD_CHECK_ARG(comm, SEMIPD_EINVAL, "ar_set_cu_trace: null communicator");
  static_cast<ArComm*>(comm)->peers.cu_trace = buf;
  return 0;
}

int semipd_ar_dispose(void* comm) {
  delete static_cast<ArComm*>(comm);
  return 0;
}

}  // extern "C"
