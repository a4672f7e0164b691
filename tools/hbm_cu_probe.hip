// How many bytes per second does ONE compute unit pull from HBM, as a function of (a) how many CUs stream at the same time,
// (b) how many waves of the CU issue loads, (c) how many bytes each wave keeps in flight, (d) register loads or LDS-DMA,
// (e) where the CUs of a share sit (spread over all 8 XCDs or packed into few)?
//
// Standalone probe (not part of libsemipd_hip.so):   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_cu_probe tools/hbm_cu_probe.hip
// Every workgroup owns its CU (100 KB of LDS), reads its own contiguous 4 MB and nothing is read twice within 2 GB (beyond
// the 256 MB Infinity Cache).  The CU set is chosen with hipExtStreamCreateWithCUMask: KFD mask bit i = XCD i % 8.
// Settles the "~36 GB/s per CU" question of DESIGN.md 3.6 against MI355X_MICROARCH.md's ldsdma-fill row.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// register loads: DEPTH independent 1 KB wave loads in flight, next set issued before the current one is consumed
template <int WAVES, int DEPTH>
__global__ void __launch_bounds__(64 * WAVES) read_regs(const char* __restrict__ base, size_t span, size_t bytes_per_wg,
                                                        size_t rot, uint32_t* __restrict__ sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wave = bytes_per_wg / WAVES;
  size_t off = ((size_t)blockIdx.x * bytes_per_wg + rot) % span + (size_t)wave * per_wave;
  const char* p = base + off + lane * 16;
  const int iters = (int)(per_wave / (DEPTH * 1024));
  u32x4 a[DEPTH], acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) a[i] = __builtin_nontemporal_load((const u32x4*)(p + i * 1024));
  for (int it = 1; it < iters; ++it) {
    p += DEPTH * 1024;
    u32x4 b[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) b[i] = __builtin_nontemporal_load((const u32x4*)(p + i * 1024));
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) acc ^= a[i];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) a[i] = b[i];
  }
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) acc ^= a[i];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345677u) sink[threadIdx.x] = acc[0];
  if (threadIdx.x == 9999) smem[0] = 1;
}

// LDS-DMA: every wave owns a ring of R slots of BLK KB and keeps R - 1 slots in flight; nothing consumes the data
template <int WAVES, int BLK, int R, int AUX>
__global__ void __launch_bounds__(64 * WAVES) read_lds(const char* __restrict__ base, size_t span, size_t bytes_per_wg,
                                                       size_t rot, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t per_wave = bytes_per_wg / WAVES;
  size_t off = ((size_t)blockIdx.x * bytes_per_wg + rot) % span + (size_t)wave * per_wave;
  const char* p = base + off + lane * 16;
  char* ring = smem + wave * (R * BLK * 1024);
  const int nblk = (int)(per_wave / (BLK * 1024));
  auto issue = [&](int b) __attribute__((always_inline)) {
    const int slot = b % R;
#pragma unroll
    for (int i = 0; i < BLK; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)b * BLK * 1024 + i * 1024),
                                       (__attribute__((address_space(3))) void*)(ring + slot * BLK * 1024 + i * 1024), 16, 0, AUX);
  };
#pragma unroll
  for (int b = 0; b < R - 1; ++b) issue(b);
  for (int b = 0; b < nblk; ++b) {
    if (b + R - 1 < nblk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * BLK) : "memory");
      issue(b + R - 1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (sink == (uint32_t*)1) sink[threadIdx.x] = smem[lane];
}

// The weight-streaming GEMM's loop taken apart (csrc/stream_linear.hip): the same per-wave LDS-DMA rings as read_lds<8, 4, 3>,
// plus, per 4 KB block: BAR = one s_barrier of the 8 waves (the shared activation ring's synchronisation), RD = that many
// ds_read_b128 sweeps of 4 KB by every wave (the GEMM reads its own 4 KB of weights once and MT x 4 KB of activations),
// MF = that many v_mfma_f32_16x16x32_bf16 on what was read.  Which of them takes the stream from ~50 to ~35 GB/s per CU?
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int WAVES, int BLK, int R, int BAR, int RD, int MF, int XKB>
__global__ void __launch_bounds__(64 * WAVES) read_anat(const char* __restrict__ base, size_t span, size_t bytes_per_wg,
                                                        size_t rot, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t per_wave = bytes_per_wg / WAVES;
  size_t off = ((size_t)blockIdx.x * bytes_per_wg + rot) % span + (size_t)wave * per_wave;
  const char* p = base + off + lane * 16;
  // XKB > 0: a shared "activation" ring of R slots of XKB KB, every wave DMAs its share (from a 64 KB region that stays in L2)
  char* xring = smem;
  char* ring = smem + R * XKB * 1024 + wave * (R * BLK * 1024);
  const char* xsrc = base + ((size_t)blockIdx.x % 64) * 65536 + lane * 16;
  constexpr int XPW = XKB > 0 ? (XKB + WAVES - 1) / WAVES : 0;   // x pieces (1 KB) per wave per block
  const int nblk = (int)(per_wave / (BLK * 1024));
  auto issue = [&](int b) __attribute__((always_inline)) {
    const int slot = b % R;
#pragma unroll
    for (int e = 0; e < XPW; ++e) {
      const int piece = (wave + e * WAVES) % (XKB > 0 ? XKB : 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + ((size_t)(b & 15) * XKB + piece) * 1024 % 65536),
                                       (__attribute__((address_space(3))) void*)(xring + slot * XKB * 1024 + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BLK; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)b * BLK * 1024 + i * 1024),
                                       (__attribute__((address_space(3))) void*)(ring + slot * BLK * 1024 + i * 1024), 16, 0, 2);
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  uint4 keep = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int b = 0; b < R - 1; ++b) issue(b);
  for (int b = 0; b < nblk; ++b) {
    if (b + R - 1 <= nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (BLK + XPW)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (b + R - 1 < nblk) issue(b + R - 1);
    const int slot = b % R;
    const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(ring + slot * BLK * 1024) + lane * 16;
    const uint32_t x0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(xring + slot * XKB * 1024) + lane * 16;
#pragma unroll
    for (int r = 0; r < RD; ++r) {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // sweep 0 reads the wave's own weights, the following sweeps the shared activation block (or the weights again)
        const uint32_t addr = (r == 0 || XKB == 0) ? a0 + i * 1024 : x0 + ((r - 1) * 4 + i) % XKB * 1024;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(addr));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MF > 0 && r * 4 + i < MF) {
          union { uint4 u; bf16x8 b; } fa, fb;
          fa.u = v[i]; fb.u = v[(i + 1) & 3];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.b, fb.b, acc, 0, 0, 0);
        } else {
          keep.x ^= v[i].x;
        }
      }
    }
  }
  if (keep.x == 0x12345677u && acc[0] == 3.f) sink[threadIdx.x] = keep.x;
}

// The weight-streaming GEMM's ACCESS PATTERN without its arithmetic: a workgroup of WAVES waves reads 16 WAVES rows of a
// row-major matrix with `pitch` bytes per row, wave w rows 16 w .. 16 w + 15, 256 bytes of each row per block (one block =
// 4 DMA instructions of 4 rows x 256 B), ring of R blocks per wave.  bytes_per_wg / (16 WAVES pitch) row groups per workgroup.
template <int WAVES, int R>
__global__ void __launch_bounds__(64 * WAVES) read_rows(const char* __restrict__ base, size_t span, size_t bytes_per_wg,
                                                        size_t rot, uint32_t* __restrict__ sink, int pitch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * (R * 4096);
  const size_t group_bytes = (size_t)16 * WAVES * pitch;
  const int groups = (int)(bytes_per_wg / group_bytes);
  const int nblk = pitch / 256;
  for (int g = 0; g < groups; ++g) {
    const size_t off = ((size_t)blockIdx.x * bytes_per_wg + rot + (size_t)g * group_bytes) % span;
    const char* p = base + off + (size_t)(wave * 16 + (lane >> 4)) * pitch + (lane & 15) * 16;
    auto issue = [&](int b) __attribute__((always_inline)) {
      const int slot = b % R;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)(4 * j) * pitch + (size_t)b * 256),
                                         (__attribute__((address_space(3))) void*)(ring + slot * 4096 + j * 1024), 16, 0, 2);
    };
#pragma unroll
    for (int b = 0; b < R - 1; ++b) issue(b);
    for (int b = 0; b < nblk; ++b) {
      if (b + R - 1 < nblk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 4) : "memory");
        issue(b + R - 1);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  if (sink == (uint32_t*)1) sink[threadIdx.x] = smem[lane];
}

static std::vector<uint32_t> make_mask(int num_cus, int n, int xcds) {
  // n CUs spread over the first `xcds` XCDs (bit i -> XCD i % 8), n / xcds per XCD, lowest CU slots first
  std::vector<uint32_t> m((num_cus + 31) / 32, 0);
  int per = n / xcds, got = 0;
  for (int cu = 0; cu < num_cus / 8 && got < n; ++cu)
    for (int x = 0; x < xcds && got < n; ++x)
      if (cu < per) { int bit = cu * 8 + x; m[bit >> 5] |= 1u << (bit & 31); ++got; }
  return m;
}

template <typename F>
static double time_it(hipStream_t st, F launch, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(0); launch(1);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) launch(i + 2);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e-3 / iters;
}

int main(int argc, char** argv) {
  int dev = 0, num_cus = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t span = (size_t)4 << 30, per_wg = (size_t)4 << 20;
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&buf, span + per_wg));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 1, span + per_wg));
  const int lds = 100 * 1024;
  if (argc <= 1 || (argc > 2 && atoi(argv[1]) == 8)) printf("# device CUs %d; every workgroup alone on its CU (100 KB LDS), 4 MB contiguous per workgroup, 4 workgroups per CU per launch\n", num_cus);
  if (argc <= 1 || (argc > 2 && atoi(argv[1]) == 8)) printf("# mode waves/CU in-flight-per-wave CUs XCDs | GB/s total | GB/s per CU\n");
#define SETLDS(k) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  // hbm_cu_probe range <lo> <hi> [<lo2> <hi2> ...]: the logical CUs lo .. hi (KFD mask bits, bit i = XCD i % 8) -- e.g. the decode
  // instance's private CUs 208 .. 255 -- contiguous streaming, the GEMM's loop, and the GEMM's row-pitched access pattern
  if (argc > 3 && argv[1][0] == 'r') {
    std::vector<uint32_t> mask((num_cus + 31) / 32, 0);
    int n = 0;
    for (int a = 2; a + 1 < argc; a += 2)
      for (int i = atoi(argv[a]); i <= atoi(argv[a + 1]) && i < num_cus; ++i)
        if (!(mask[i >> 5] >> (i & 31) & 1)) { mask[i >> 5] |= 1u << (i & 31); ++n; }
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    printf("# CUs");
    for (int a = 2; a + 1 < argc; a += 2) printf(" %s-%s", argv[a], argv[a + 1]);
    printf(" (%d CUs): mode | waves in-flight-per-wave CUs - | GB/s total | GB/s per CU\n", n);
    struct { int n, xcds; } sh = {n, 0};
    const int wgs = n * 4;
    const size_t total = (size_t)wgs * per_wg;
    auto rot = [&](int i) { return ((size_t)i * total) % span; };
#define RUNP(label, waves, kernel, ldsbytes, pitch)                                                               \
    {                                                                                                              \
      SETLDS(kernel);                                                                                              \
      double t = time_it(st, [&](int i) { hipLaunchKernelGGL(kernel, dim3(wgs), dim3(64 * waves), ldsbytes, st,    \
                                                              (const char*)buf, span, per_wg, rot(i), sink, pitch); }, 10); \
      CK(hipGetLastError());                                                                                       \
      printf("%-8s | %d pitch %5d  %3d - | %7.0f | %6.1f\n", label, waves, pitch, n, total / t / 1e9, total / t / 1e9 / n); \
      fflush(stdout);                                                                                              \
    }
#define RUN(label, waves, inflight_kb, kernel, ldsbytes)                                                          \
    {                                                                                                              \
      SETLDS(kernel);                                                                                              \
      double t = time_it(st, [&](int i) { hipLaunchKernelGGL(kernel, dim3(wgs), dim3(64 * waves), ldsbytes, st,    \
                                                              (const char*)buf, span, per_wg, rot(i), sink); }, 10); \
      CK(hipGetLastError());                                                                                       \
      printf("%-8s | %d %3d KB  %3d %d | %7.0f | %6.1f\n", label, waves, inflight_kb, sh.n, sh.xcds, total / t / 1e9,      \
             total / t / 1e9 / sh.n);                                                                              \
      fflush(stdout);                                                                                              \
    }
    RUN("lds-nt", 8, 8, (read_lds<8, 4, 3, 2>), lds)
    RUN("lds-nt", 4, 24, (read_lds<4, 8, 4, 2>), 128 * 1024)
    RUN("regs-nt", 8, 8, (read_regs<8, 8>), lds)
    RUN("anat bar+x8+rd3+mf8 (= the GEMM at 32 rows)", 8, 8, (read_anat<8, 4, 3, 1, 3, 8, 8>), 128 * 1024)
    RUNP("rows r3", 8, (read_rows<8, 3>), lds, 8192)
    RUNP("rows r3", 8, (read_rows<8, 3>), lds, 28672)
    RUNP("rows r3", 8, (read_rows<8, 3>), lds, 2048)
    RUNP("rows r5", 8, (read_rows<8, 5>), 160 * 1024, 8192)
    RUNP("rows r3 4w", 4, (read_rows<4, 3>), lds, 8192)
#undef RUN
#undef RUNP
    CK(hipStreamDestroy(st));
    return 0;
  }
  struct Share { int n, xcds; };
  // one share per invocation (a hang then costs one point, not the sweep):  hbm_cu_probe <CUs> <XCDs> | hbm_cu_probe pair
  const bool pair_only = argc > 1 && argv[1][0] == 'p';
  const Share shares[] = {{argc > 2 ? atoi(argv[1]) : 96, argc > 2 ? atoi(argv[2]) : 8}};
  for (const Share& sh : shares) {
    if (sh.n > num_cus || pair_only) continue;
    std::vector<uint32_t> mask = make_mask(num_cus, sh.n, sh.xcds);
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    const int wgs = sh.n * 4;
    const size_t total = (size_t)wgs * per_wg;
    auto rot = [&](int i) { return ((size_t)i * total) % span; };
#define RUN(label, waves, inflight_kb, kernel, ldsbytes)                                                          \
    {                                                                                                              \
      SETLDS(kernel);                                                                                              \
      double t = time_it(st, [&](int i) { hipLaunchKernelGGL(kernel, dim3(wgs), dim3(64 * waves), ldsbytes, st,    \
                                                              (const char*)buf, span, per_wg, rot(i), sink); }, 10); \
      CK(hipGetLastError());                                                                                       \
      printf("%-8s | %d %3d KB  %3d %d | %7.0f | %6.1f\n", label, waves, inflight_kb, sh.n, sh.xcds, total / t / 1e9,      \
             total / t / 1e9 / sh.n);                                                                              \
      fflush(stdout);                                                                                              \
    }
    // LDS-DMA, nt (aux 2): waves x ring depth
    RUN("lds-nt", 1, 12, (read_lds<1, 4, 4, 2>), lds)
    RUN("lds-nt", 1, 48, (read_lds<1, 16, 4, 2>), lds)
    RUN("lds-nt", 2, 12, (read_lds<2, 4, 4, 2>), lds)
    RUN("lds-nt", 4, 12, (read_lds<4, 4, 4, 2>), lds)
    RUN("lds-nt", 8, 8, (read_lds<8, 4, 3, 2>), lds)
    RUN("lds-nt", 8, 12, (read_lds<8, 4, 4, 2>), 128 * 1024)
    RUN("lds-nt", 4, 24, (read_lds<4, 8, 4, 2>), 128 * 1024)
    RUN("lds", 8, 8, (read_lds<8, 4, 3, 0>), lds)
    // register loads
    RUN("regs-nt", 1, 8, (read_regs<1, 8>), lds)
    RUN("regs-nt", 4, 8, (read_regs<4, 8>), lds)
    RUN("regs-nt", 8, 8, (read_regs<8, 8>), lds)
    RUN("regs-nt", 8, 16, (read_regs<8, 16>), lds)
    RUN("regs-nt", 16, 8, (read_regs<16, 8>), lds)
    // the GEMM loop taken apart: 8 waves x ring of 3 x 4 KB, + barrier, + LDS sweeps, + MFMAs, + shared 8 KB activation ring
    RUN("anat --------", 8, 8, (read_anat<8, 4, 3, 0, 0, 0, 0>), lds)
    RUN("anat bar", 8, 8, (read_anat<8, 4, 3, 1, 0, 0, 0>), lds)
    RUN("anat rd1", 8, 8, (read_anat<8, 4, 3, 0, 1, 0, 0>), lds)
    RUN("anat rd3", 8, 8, (read_anat<8, 4, 3, 0, 3, 0, 0>), lds)
    RUN("anat bar+rd3", 8, 8, (read_anat<8, 4, 3, 1, 3, 0, 0>), lds)
    RUN("anat bar+rd3+mf8", 8, 8, (read_anat<8, 4, 3, 1, 3, 8, 0>), lds)
    RUN("anat bar+x8", 8, 8, (read_anat<8, 4, 3, 1, 0, 0, 8>), 128 * 1024)
    RUN("anat bar+x8+rd3", 8, 8, (read_anat<8, 4, 3, 1, 3, 0, 8>), 128 * 1024)
    RUN("anat bar+x8+rd3+mf8 (= the GEMM at 32 rows)", 8, 8, (read_anat<8, 4, 3, 1, 3, 8, 8>), 128 * 1024)
    RUN("anat bar+x8+rd2+mf8 (k-split waves)", 8, 8, (read_anat<8, 4, 3, 1, 2, 8, 8>), 128 * 1024)
    RUN("anat bar+x16+rd5+mf16 (64 rows)", 8, 8, (read_anat<8, 4, 3, 1, 5, 16, 16>), 150 * 1024)
    RUN("anat ring4 bar+x8+rd3+mf8", 8, 12, (read_anat<8, 4, 4, 1, 3, 8, 8>), 160 * 1024)
    RUN("anat 4 waves x 8 KB blocks bar+x8+rd3+mf8", 4, 16, (read_anat<4, 8, 3, 1, 3, 8, 8>), 128 * 1024)
    CK(hipStreamDestroy(st));
  }
  // two disjoint shares streaming at the same time (decode-like 96 from the top, a second stream on the other 160)
  if (pair_only) {
    std::vector<uint32_t> lo((num_cus + 31) / 32, 0), hi((num_cus + 31) / 32, 0);
    for (int i = 0; i < 160; ++i) lo[i >> 5] |= 1u << (i & 31);
    for (int i = 160; i < 256 && i < num_cus; ++i) hi[i >> 5] |= 1u << (i & 31);
    hipStream_t s_lo, s_hi;
    CK(hipExtStreamCreateWithCUMask(&s_lo, (uint32_t)lo.size(), lo.data()));
    CK(hipExtStreamCreateWithCUMask(&s_hi, (uint32_t)hi.size(), hi.data()));
    SETLDS((read_lds<8, 4, 3, 2>));
    const int wg_hi = 96 * 4, wg_lo = 160 * 4;
    // keep the low share busy for the whole measurement of the high share
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 40; ++i)
        hipLaunchKernelGGL((read_lds<8, 4, 3, 2>), dim3(wg_lo), dim3(512), lds, s_lo, (const char*)buf, span, per_wg,
                           ((size_t)i * wg_lo * per_wg + ((size_t)2 << 30)) % span, sink);
      double t = time_it(s_hi, [&](int i) { hipLaunchKernelGGL((read_lds<8, 4, 3, 2>), dim3(wg_hi), dim3(512), lds, s_hi,
                                                               (const char*)buf, span, per_wg, ((size_t)i * wg_hi * per_wg) % span, sink); }, 10);
      CK(hipDeviceSynchronize());
      printf("96-CU share streaming NEXT TO a streaming 160-CU share: %7.0f GB/s = %5.1f per CU\n", wg_hi * (double)per_wg / t / 1e9,
             wg_hi * (double)per_wg / t / 1e9 / 96);
    }
  }
  return 0;
}
