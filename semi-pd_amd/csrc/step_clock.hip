// Decode-step deadline gate: the prefill instance yields at a layer boundary while a decode step is overdue.
//
// What it stands in for: the reference sizes the two instances' MPS percentages so that decode latency holds (semi_pd/
// utils.py:10-11, entrypoints/engine.py:588-634: prefill 80 %, decode 100 %); the percentages are static, so the decode
// tail under a prefill batch is whatever the private 20 % deliver.  On MI355X a decode step next to a prefill batch runs
// mostly on the CUs the prefill share leaves free -- 32 CUs next to a 224-CU prefill share stream 1.8 TB/s, a step of 19 GB
// then takes 11-15 ms against 4.5 alone -- and the share that makes prefill fast (224 CUs: TTFT p50 29 ms) is the one
// that makes this tail long (TBT p99 15.5 ms; profiles/r04_policy_sweep_3.txt).  The gate bounds the tail instead of
// paying for it with CUs all the time:
//   * the decode instance stamps a slot in device memory both instances map (uncached, system-scope atomics): the first
//     node of every decode step writes the device's wall clock, the last node writes 0;
//   * the prefill instance launches step_clock_gate_kernel (one wave) between decoder layers on its compute stream: if a
//     decode step is in flight and older than the deadline, the wave sleeps until the stamp changes -- the stream behind
//     it holds, the prefill share drains (one kernel, < 0.4 ms), the decode step finishes on the whole chip -- or until
//     max_wait has passed (a decode instance that died must not hang its neighbour).
// Nothing on the host is involved: both sides are kernels in stream order, capturable in a hipGraph.
#include "common.h"

namespace semipd {

// slot layout (uint64 words): [0] start stamp of the step in flight (0 = none), [1] steps begun, [2..7] reserved
// stats layout (uint64 words, the gating process's own memory): [0] gates passed, [1] gates that held, [2] ticks held,
// [3] holds that ended by max_wait
__global__ void step_clock_mark_kernel(uint64_t* slot, int begin) {
  if (threadIdx.x != 0) return;
  if (begin) {
    uint64_t now = wall_clock64();
    if (now == 0) now = 1;
    __hip_atomic_store(slot + 1, __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot, now, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    __hip_atomic_store(slot, (uint64_t)0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void step_clock_gate_kernel(const uint64_t* slot, uint64_t deadline_ticks, uint64_t max_wait_ticks,
                                       uint64_t* stats) {
  if (threadIdx.x != 0) return;
  stats[0] += 1;
  const uint64_t s = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (s == 0) return;
  const uint64_t t0 = wall_clock64();
  if (t0 - s < deadline_ticks || t0 < s) return;   // in flight, not overdue (or a stamp from the future: ignore)
  stats[1] += 1;
  uint64_t now = t0;
  while (__hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == s) {
    __builtin_amdgcn_s_sleep(32);
    now = wall_clock64();
    if (now - t0 > max_wait_ticks) {
      stats[3] += 1;
      break;
    }
  }
  stats[2] += now - t0;
}

}  // namespace semipd

using namespace semipd;

extern "C" {

int semipd_step_clock_mark(void* slot, int begin, void* stream) {
  SEMIPD_CHECK_ARG(slot && (reinterpret_cast<uintptr_t>(slot) & 7u) == 0, SEMIPD_EINVAL, "step_clock_mark: null or unaligned slot");
  hipLaunchKernelGGL(step_clock_mark_kernel, dim3(1), dim3(64), 0, as_stream(stream), (uint64_t*)slot, begin ? 1 : 0);
  return launch_status("step_clock_mark");
}

int semipd_step_clock_gate(const void* slot, uint64_t deadline_ticks, uint64_t max_wait_ticks, void* stats, void* stream) {
  SEMIPD_CHECK_ARG(slot && stats && (reinterpret_cast<uintptr_t>(slot) & 7u) == 0 && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0,
                   SEMIPD_EINVAL, "step_clock_gate: null or unaligned pointer");
  SEMIPD_CHECK_ARG(max_wait_ticks > 0 && max_wait_ticks <= (uint64_t)1 << 32, SEMIPD_EINVAL,
                   "step_clock_gate: max_wait must be positive and finite -- a gate must never hold a stream for good");
  hipLaunchKernelGGL(step_clock_gate_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const uint64_t*)slot, deadline_ticks,
                     max_wait_ticks, (uint64_t*)stats);
  return launch_status("step_clock_gate");
}

/* ticks per millisecond of the clock the stamps are taken from (wall_clock64: hipDeviceAttributeWallClockRate is in kHz;
 * 100 MHz on gfx950) */
int semipd_step_clock_ticks_per_ms(int device, uint64_t* ticks) {
  SEMIPD_CHECK_ARG(ticks, SEMIPD_EINVAL, "step_clock_ticks_per_ms: null pointer");
  int khz = 0;
  SEMIPD_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
  *ticks = khz > 0 ? (uint64_t)khz : 100000;
  return 0;
}

}  // extern "C"
