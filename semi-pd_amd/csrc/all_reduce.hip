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

static size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

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

int semipd_ar_dispose(void* comm) {
  delete static_cast<ArComm*>(comm);
  return 0;
}

}  // extern "C"
