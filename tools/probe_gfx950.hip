// Hardware probe (run on the GPU box): prints the lane/element mapping of ds_read_b64_tr_b16
// so LDS layouts for transposed MFMA operands can be derived from measurement, not guesswork.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void tr_probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // mode 0: lane l points at elements [4l, 4l+4)
  // mode 1: 16-lane group g reads a [4 rows][16 cols] block at element offset g*64: lane i points at row (i>>2), col 4*(i&3)
  int elem = mode == 0 ? 4 * l : (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[0]) + elem * 2;
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[l * 4 + 0] = r.x & 0xffff;
  out[l * 4 + 1] = r.x >> 16;
  out[l * 4 + 2] = r.y & 0xffff;
  out[l * 4 + 3] = r.y >> 16;
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 256 * 2);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d, mode);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16 mode %d (lane: 4 element indices returned)\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz memclk %d kHz L2 %d B gcn %s\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize, p.gcnArchName);
  return 0;
}
