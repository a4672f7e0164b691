// semi-pd-ipc for ROCm: hipIpcMemHandle export/import with a per-process mapping cache,
// CU-count query and CU-masked stream creation.
//
// Differences from the reference module (semi-pd-ipc/ipc.cpp:60-97), on purpose:
//  * the handle is always taken on the *allocation base* (hipMemGetAddressRange), and the
//    byte offset of the tensor inside it is returned alongside, so callers do not need
//    torch's storage()._share_cuda_() side effects (semi_pd/utils.py:66-76);
//  * hipIpcOpenMemHandle may be called only once per allocation per process, while PyTorch's
//    caching allocator packs hundreds of tensors into one allocation: opens are cached by
//    handle bytes and reference counted, close unmaps at zero (the reference never closes);
//  * errors are returned, never exit(1) (ipc.cpp:74-79).
#include "common.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>

namespace semipd {

struct Mapping {
  void* base;
  int device;
  int refs;
};
static std::mutex g_ipc_mu;
static std::map<std::string, Mapping> g_by_handle;
static std::map<void*, std::string> g_by_base;

__global__ void probe_cu_placement_kernel(int32_t* out, long long spin_cycles) {
  if (threadIdx.x == 0) {
    // HW_REG_XCC_ID (id 20): bits [3:0] = XCC id.  HW_REG_HW_ID (id 4): CU id bits [11:8],
    // SH id bit 12, SE id bits [15:13].
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf;
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    out[2 * blockIdx.x] = (int32_t)xcc;
    out[2 * blockIdx.x + 1] = (int32_t)((hw >> 8) & 0xff);  // cu | sh<<4 | se<<5
  }
  const long long t0 = clock64();
  while (clock64() - t0 < spin_cycles) {
  }
}

}  // namespace semipd

using namespace semipd;

namespace semipd {
__global__ void noop_kernel() {}
}  // namespace semipd

extern "C" {

int semipd_ipc_get_handle(const void* dev_ptr, uint8_t handle[64], uint64_t* offset) {
  SEMIPD_CHECK_ARG(dev_ptr && handle && offset, SEMIPD_EINVAL, "ipc_get_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t must be 64 bytes");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  SEMIPD_HIP(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr));
  // ROCm 7.2 / dmabuf IPC: hipIpcOpenMemHandle in the importing process never returns when the
  // allocation size modulo 4 GiB is 2 GiB or more (measured: 2.5 GiB and 6.1 GiB hang; 1.9, 5.0, 9.5, 20,
  // 40 GiB map in < 1 ms; tools/ipc_big_probe.py).  Refuse to export such an allocation instead of
  // hanging the importer; large buffers meant for sharing are sized around it (memory_pool.py).
  // The rule was measured on ONE runtime (HIP 7.2): it is applied on every runtime (padding is harmless where the
  // importer would not have hung), can be switched off with SEMIPD_IPC_SIZE_RULE=off once a runtime is known to be
  // fixed, and the importer bounds every hipIpcOpenMemHandle with a watchdog (semi_pd_ipc.py) so that a size this
  // rule does not know about fails with a message instead of hanging the prefill instance.
  static const bool rule_off = [] { const char* e = getenv("SEMIPD_IPC_SIZE_RULE"); return e && std::string(e) == "off"; }();
  SEMIPD_CHECK_ARG(rule_off || ((uint64_t)size & 0xffffffffull) < 0x80000000ull, SEMIPD_EINVAL,
                   "ipc_get_handle: allocation of %zu bytes cannot be imported by another process on this ROCm "
                   "(size mod 4 GiB >= 2 GiB hangs hipIpcOpenMemHandle); allocate it with ipc-safe padding",
                   size);
  hipIpcMemHandle_t h;
  memset(&h, 0, sizeof(h));  // the runtime fills only part of the 64 bytes; the importer's mapping cache
                             // is keyed by all of them, so the rest must not be stack garbage
  SEMIPD_HIP(hipIpcGetMemHandle(&h, base));
  memcpy(handle, &h, 64);
  *offset = (uint64_t)((const uint8_t*)dev_ptr - (const uint8_t*)base);
  return 0;
}

int semipd_runtime_version(int* runtime, int* driver) {
  SEMIPD_CHECK_ARG(runtime && driver, SEMIPD_EINVAL, "runtime_version: null pointer");
  SEMIPD_HIP(hipRuntimeGetVersion(runtime));
  SEMIPD_HIP(hipDriverGetVersion(driver));
  return 0;
}

int semipd_stream_create(int device, void** stream) {
  SEMIPD_CHECK_ARG(stream, SEMIPD_EINVAL, "stream_create: null pointer");
  int prev = -1;
  SEMIPD_HIP(hipGetDevice(&prev));
  if (device >= 0 && device != prev) SEMIPD_HIP(hipSetDevice(device));
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (device >= 0 && device != prev) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  *stream = (void*)s;
  return 0;
}

int semipd_stream_abort_capture(void* stream) {
  SEMIPD_CHECK_ARG(stream, SEMIPD_EINVAL, "stream_abort_capture: null stream");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipGraph_t graph = nullptr;
  (void)hipStreamEndCapture(st, &graph);   // returns hipErrorStreamCaptureInvalidated and leaves the stream invalidated
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(st);              // the stream takes the invalidated capture with it
  for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; ++i) {}
  return 0;
}

int semipd_clear_last_error(void) {
  int n = 0;
  while (n < 8 && hipGetLastError() != hipSuccess) ++n;
  return n;
}

int semipd_ipc_open(const uint8_t handle[64], int device, void** base) {
  SEMIPD_CHECK_ARG(handle && base, SEMIPD_EINVAL, "ipc_open: null pointer");
  const std::string key((const char*)handle, 64);
  std::lock_guard<std::mutex> g(g_ipc_mu);
  auto it = g_by_handle.find(key);
  if (it != g_by_handle.end()) {
    it->second.refs += 1;
    *base = it->second.base;
    return 0;
  }
  int prev = -1;
  SEMIPD_HIP(hipGetDevice(&prev));
  if (device >= 0 && device != prev) SEMIPD_HIP(hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (device >= 0 && device != prev) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return (int)e;
  }
  g_by_handle[key] = Mapping{p, device, 1};
  g_by_base[p] = key;
  *base = p;
  return 0;
}

int semipd_ipc_close(void* base) {
  std::lock_guard<std::mutex> g(g_ipc_mu);
  auto it = g_by_base.find(base);
  SEMIPD_CHECK_ARG(it != g_by_base.end(), SEMIPD_ENOTFOUND, "ipc_close: %p is not an open mapping",
                   base);
  Mapping& m = g_by_handle[it->second];
  if (--m.refs > 0) return 0;
  hipError_t e = hipIpcCloseMemHandle(base);
  g_by_handle.erase(it->second);
  g_by_base.erase(it);
  if (e != hipSuccess) {
    set_error("hipIpcCloseMemHandle failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

int semipd_ipc_num_open(void) {
  std::lock_guard<std::mutex> g(g_ipc_mu);
  return (int)g_by_handle.size();
}

int semipd_device_cu_count(int device, int* num_cus) {
  SEMIPD_CHECK_ARG(num_cus, SEMIPD_EINVAL, "device_cu_count: null pointer");
  SEMIPD_HIP(hipSetDevice(device));  // same side effect as GetDeviceSMCount (ipc.cpp:88)
  int n = 0;
  SEMIPD_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
  *num_cus = n;
  return 0;
}

int semipd_cu_mask_fill(int num_cus, int percent, int from_top, uint32_t* mask, int words) {
  SEMIPD_CHECK_ARG(mask && num_cus > 0 && words * 32 >= num_cus && percent > 0 && percent <= 100,
                   SEMIPD_EINVAL, "cu_mask_fill: bad arguments");
  // The KFD spreads consecutive mask bits round-robin over the XCDs (and shader engines inside
  // an XCD), so a contiguous range of logical CU bits is automatically XCD-balanced: bit i lands
  // on XCD i % 8.  We enable round(num_cus*percent/100) bits, rounded to a multiple of 8 so every
  // XCD gets the same number of CUs, from the bottom or from the top of the range.  A share taken from the TOP is what
  // the complementary share from the bottom leaves (num_cus - bottom(100 - percent)): the two roundings of a pair like
  // 62 / 38 can then never claim the same group (on 304 CUs they did: 192 + 120), and on 256 CUs nothing changes for
  // the pairs in use (160 / 96, 128 / 128, 192 / 64).
  // Granule (round 5): 32 logical CUs = one per shader engine of every XCD (8 XCDs x 4 SEs) wherever the device's CU count
  // allows it.  The dispatcher deals a kernel's workgroups to the shader engines round-robin, not to whichever CU is
  // free, so a share with 6 CUs per XCD (two SEs with 2, two with 1) runs at the pace of its one-CU engines: measured,
  // any 48 CUs stream 1.82 TB/s where 32 stream 1.83 and 64 stream 3.4 (profiles/r05_hbm_probe_cu_ranges.txt), and
  // prefill shares of 208 / 216 CUs served exactly like 192, 232 / 240 exactly like 224 (profiles/r04_policy_sweep_3.txt).
  // 80 % of 256 CUs is therefore 192, not 208: the 16 CUs in between were held by the prefill instance without making it
  // faster and are worth a second private CU per shader engine to the decode instance.
  const int g = (num_cus % 32 == 0 && num_cus >= 256) ? 32 : 8;   // (8 XCDs x 4 SEs: the MI300 / MI355 layout)
  auto bottom = [num_cus, g](int pct) {
    int m = (num_cus * pct + 50) / 100;
    m = (m + g / 2) / g * g;
    return m > num_cus ? num_cus : m;
  };
  int n = from_top ? num_cus - bottom(100 - percent) : bottom(percent);
  if (n < g) n = g;
  if (n > num_cus) n = num_cus;
  for (int w = 0; w < words; ++w) mask[w] = 0;
  const int lo = from_top ? num_cus - n : 0;
  for (int i = lo; i < lo + n; ++i) mask[i >> 5] |= (1u << (i & 31));
  return n;
}

int semipd_stream_create_cu_mask(int device, const uint32_t* mask, int words, void** stream) {
  SEMIPD_CHECK_ARG(mask && stream && words > 0, SEMIPD_EINVAL, "stream_create_cu_mask: bad arguments");
  int prev = -1;
  SEMIPD_HIP(hipGetDevice(&prev));
  if (device >= 0 && device != prev) SEMIPD_HIP(hipSetDevice(device));
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (device >= 0 && device != prev) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  *stream = (void*)s;
  return 0;
}

int semipd_stream_create_with_priority(int device, int priority, void** stream, int* range) {
  SEMIPD_CHECK_ARG(stream, SEMIPD_EINVAL, "stream_create_with_priority: null pointer");
  int prev = -1;
  SEMIPD_HIP(hipGetDevice(&prev));
  if (device >= 0 && device != prev) SEMIPD_HIP(hipSetDevice(device));
  int least = 0, greatest = 0;   // numerically: least >= greatest
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipStream_t s = nullptr;
  if (e == hipSuccess) {
    if (range) range[0] = least, range[1] = greatest;
    const int p = priority > least ? least : (priority < greatest ? greatest : priority);
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p);
  }
  if (device >= 0 && device != prev) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("hipStreamCreateWithPriority(%d) failed: %s", priority, hipGetErrorString(e));
    (void)hipGetLastError();
    return (int)e;
  }
  *stream = (void*)s;
  return 0;
}

int semipd_stream_destroy(void* stream) {
  SEMIPD_CHECK_ARG(stream, SEMIPD_EINVAL, "stream_destroy: null stream");
  SEMIPD_HIP(hipStreamDestroy(as_stream(stream)));
  return 0;
}

int semipd_stream_get_cu_mask(void* stream, uint32_t* mask, int words) {
  SEMIPD_CHECK_ARG(mask && words > 0, SEMIPD_EINVAL, "stream_get_cu_mask: bad arguments");
  SEMIPD_HIP(hipExtStreamGetCUMask(as_stream(stream), (uint32_t)words, mask));
  return 0;
}

int semipd_probe_cu_placement(int32_t* out, int num_workgroups, int64_t spin_cycles, void* stream) {
  SEMIPD_CHECK_ARG(out && num_workgroups > 0, SEMIPD_EINVAL, "probe_cu_placement: bad arguments");
  hipLaunchKernelGGL(probe_cu_placement_kernel, dim3(num_workgroups), dim3(64), 0, as_stream(stream),
                     out, (long long)spin_cycles);
  return launch_status("probe_cu_placement");
}

// ---- share board: one page of host memory both instances of a GPU map (a file in the engine's socket directory) ----
// 64 slots of one 64-bit word, each on its own cache line is not needed: a slot has ONE writer and is read a few hundred
// times per second.  The words are std::atomic on a MAP_SHARED page: lock-free 8-byte atomics work across processes.
enum { kBoardSlots = 64, kBoardBytes = 4096 };
static_assert(sizeof(std::atomic<int64_t>) == 8 && std::atomic<int64_t>::is_always_lock_free, "8-byte lock-free atomics");

int semipd_share_board_open(const char* path, int create, void** board) {
  SEMIPD_CHECK_ARG(path && board, SEMIPD_EINVAL, "share_board_open: null pointer");
  const int fd = open(path, create ? (O_RDWR | O_CREAT) : O_RDWR, 0600);
  if (fd < 0) {
    set_error("share_board_open: cannot open %s: %s", path, strerror(errno));
    return SEMIPD_ENOTFOUND;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || (sb.st_size < kBoardBytes && (!create || ftruncate(fd, kBoardBytes) != 0))) {
    // (a fresh file is zero-filled by ftruncate: every slot starts at 0 = "idle")
    set_error("share_board_open: %s is not a board (size %lld) and cannot be sized: %s", path, (long long)sb.st_size,
              strerror(errno));
    close(fd);
    return SEMIPD_EINVAL;
  }
  void* p = mmap(nullptr, kBoardBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    set_error("share_board_open: mmap of %s failed: %s", path, strerror(errno));
    return SEMIPD_EINVAL;
  }
  *board = p;
  return 0;
}

int semipd_share_board_close(void* board) {
  SEMIPD_CHECK_ARG(board, SEMIPD_EINVAL, "share_board_close: null board");
  munmap(board, kBoardBytes);
  return 0;
}

int semipd_share_board_store(void* board, int slot, int64_t value) {
  SEMIPD_CHECK_ARG(board && slot >= 0 && slot < kBoardSlots, SEMIPD_EINVAL, "share_board_store: bad slot %d", slot);
  reinterpret_cast<std::atomic<int64_t>*>(board)[slot].store(value, std::memory_order_release);
  return 0;
}

int semipd_share_board_add(void* board, int slot, int64_t delta, int64_t* result) {
  SEMIPD_CHECK_ARG(board && slot >= 0 && slot < kBoardSlots, SEMIPD_EINVAL, "share_board_add: bad slot %d", slot);
  const int64_t v = reinterpret_cast<std::atomic<int64_t>*>(board)[slot].fetch_add(delta, std::memory_order_acq_rel) + delta;
  if (result) *result = v;
  return 0;
}

int semipd_share_board_load(void* board, int slot, int64_t* value) {
  SEMIPD_CHECK_ARG(board && value && slot >= 0 && slot < kBoardSlots, SEMIPD_EINVAL, "share_board_load: bad slot %d", slot);
  *value = reinterpret_cast<std::atomic<int64_t>*>(board)[slot].load(std::memory_order_acquire);
  return 0;
}

int semipd_launch_noop(int count, void* stream) {
  SEMIPD_CHECK_ARG(count >= 0 && count <= 1024, SEMIPD_EINVAL, "launch_noop: bad count");
  for (int i = 0; i < count; ++i) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, as_stream(stream));
  return launch_status("noop");
}

}  // extern "C"
