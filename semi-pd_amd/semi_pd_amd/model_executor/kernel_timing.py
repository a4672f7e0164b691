"""Live per-kernel timing with HIP events on the launching stream (bench.py's roofline numbers).

A sampled decode step runs eagerly (not through the hipGraph) and brackets every decode-attention
launch with a pair of events recorded on the stream the kernel is launched on; prefill batches
bracket the extend-attention launches the same way.  Alongside the elapsed time we keep the
ALGORITHMIC bytes / flops of each launch (SURVEY §8d formulas) so that
achieved = sum(bytes) / sum(time) is independent of how many launches were sampled.

Two corrections make the event pair measure the kernel rather than the launch path:
  * a short device-side spin in front of the start event (an eager step is CPU-bound: without it the
    start event retires before the kernel has even been submitted);
  * the dispatch + completion overhead an event pair adds around k back-to-back kernels is calibrated
    once with k EMPTY kernels (semipd_launch_noop) under the same conditions (incl. a profiler, if
    one is attached).  The headline rate uses the raw interval (conservative); the interval minus the
    empty-kernel overhead is reported next to it.  rocprofv3's per-kernel duration (begin -> end of the
    dispatch, no queue overhead) lies between the two."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Tuple

import torch


class KernelTiming:
    def __init__(self, sample_every: int = 16, max_pending: int = 4096):
        self.sample_every = sample_every
        self.max_pending = max_pending
        self.gate_cycles = 0  # calibrated on first use (see start())
        self.overhead_ms: Dict[int, float] = {}
        self._step = 0
        self.active = False
        self.linear_budget = 0   # streaming-GEMM launches still to be timed in this step (the first layer's four)
        self._pending: List[Tuple[str, torch.cuda.Event, torch.cuda.Event, float, float, int]] = []
        self._acc: Dict[str, List[float]] = defaultdict(lambda: [0.0, 0.0, 0.0, 0, 0.0])  # ms, bytes, flops, n, raw ms

    def begin_step(self) -> bool:
        """Called once per forward; returns True when this step is a sampled (eager, timed) one."""
        self._step += 1
        # A function of the step count ALONE.  A sampled step runs eagerly on the raw batch, the others replay a graph
        # captured for a padded batch size: under tensor parallelism every rank must take the same path for the same step,
        # or the peer-memory collectives of the two paths (different payloads -> different block counts) wait for each other
        # for ever.  (Until round 5 a full `_pending` list switched sampling off -- on the ranks that nobody asks for
        # statistics it fills up, on rank 0 it is drained: after ~8 k steps the ranks disagreed and an N = 2 run hung.)
        self.active = self._step % self.sample_every == 0
        self.linear_budget = 4 if self.active else 0
        return self.active

    def end_step(self):
        self.active = False

    @staticmethod
    def _calibrate_gate(target_us: float = 60.0) -> int:
        """Ticks of torch.cuda._sleep that spin for ~target_us (the tick unit differs per platform)."""
        probe = 20_000
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(probe)
        torch.cuda.synchronize()
        s.record()
        torch.cuda._sleep(probe)
        e.record()
        torch.cuda.synchronize()
        us = max(s.elapsed_time(e) * 1e3, 1.0)
        return max(int(probe * target_us / us), 1)

    def _calibrate_overhead(self, n_kernels: int, reps: int = 40) -> float:
        """Median elapsed time of [gate, start event, n empty kernels, stop event] in ms."""
        from semi_pd_amd import _lib
        lib = _lib.load()
        stream = torch.cuda.current_stream()
        vals = []
        for _ in range(reps):
            torch.cuda._sleep(self.gate_cycles)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            _lib.check(lib.semipd_launch_noop(n_kernels, stream.cuda_stream), "launch_noop")
            e.record(stream)
            vals.append((s, e))
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in vals)
        return ms[len(ms) // 2]

    def start(self) -> torch.cuda.Event:
        if self.gate_cycles == 0:
            self.gate_cycles = self._calibrate_gate()
        torch.cuda._sleep(self.gate_cycles)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def stop(self, name: str, start: torch.cuda.Event, nbytes: float, flops: float = 0.0, n_kernels: int = 1):
        if len(self._pending) >= self.max_pending:
            return                     # nobody has collected for a long time: this sample is dropped (the step stays a sampled one)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self._pending.append((name, start, e, float(nbytes), float(flops), int(n_kernels)))

    def _drain(self):
        if not self._pending:
            return
        torch.cuda.synchronize()
        for name, s, e, b, f, k in self._pending:
            if k not in self.overhead_ms:
                self.overhead_ms[k] = self._calibrate_overhead(k)
            raw = s.elapsed_time(e)
            a = self._acc[name]
            a[0] += max(raw - self.overhead_ms[k], raw * 0.25)
            a[1] += b
            a[2] += f
            a[3] += 1
            a[4] += raw
        self._pending.clear()

    def summary(self) -> Dict[str, dict]:
        self._drain()
        out = {}
        for name, (ms, b, f, n, raw) in self._acc.items():
            if n == 0:
                continue
            # the reported rate uses the RAW event interval (dispatch + completion of the launches
            # included): a lower bound on the kernel's own rate.  avg_us_minus_event_overhead is the
            # same interval minus the calibrated empty-kernel interval: a lower bound on the duration.
            out[name] = {"launches": n, "avg_us": raw * 1e3 / n, "avg_us_minus_event_overhead": ms * 1e3 / n,
                         "bytes_per_launch": b / n, "flops_per_launch": f / n,
                         "gbps": (b / (raw * 1e-3)) / 1e9 if raw > 0 else 0.0,
                         "tflops": (f / (raw * 1e-3)) / 1e12 if raw > 0 else 0.0}
        out_over = {str(k): round(v * 1e3, 2) for k, v in self.overhead_ms.items()}
        if out and out_over:
            out["_event_pair_overhead_us"] = out_over
        return out

    def reset(self):
        self._drain()
        self._acc.clear()
