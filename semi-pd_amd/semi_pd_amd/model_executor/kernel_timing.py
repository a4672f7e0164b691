"""Live per-kernel timing with HIP events on the launching stream (bench.py's roofline numbers).

A sampled decode step runs eagerly (not through the hipGraph) and brackets every decode-attention
launch with a pair of events recorded on the stream the kernel is launched on; prefill batches
bracket the extend-attention launches the same way.  Alongside the elapsed time we keep the
ALGORITHMIC bytes / flops of each launch (SURVEY §8d formulas) so that
achieved = sum(bytes) / sum(time) is independent of how many launches were sampled."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Tuple

import torch


class KernelTiming:
    def __init__(self, sample_every: int = 16, max_pending: int = 4096):
        self.sample_every = sample_every
        self.max_pending = max_pending
        self.gate_cycles = 0  # calibrated on first use (see start())
        self._step = 0
        self.active = False
        self._pending: List[Tuple[str, torch.cuda.Event, torch.cuda.Event, float, float]] = []
        self._acc: Dict[str, List[float]] = defaultdict(lambda: [0.0, 0.0, 0.0, 0])  # ms, bytes, flops, n

    def begin_step(self) -> bool:
        """Called once per forward; returns True when this step is a sampled (eager, timed) one."""
        self._step += 1
        self.active = (self._step % self.sample_every == 0) and len(self._pending) < self.max_pending
        return self.active

    def end_step(self):
        self.active = False

    @staticmethod
    def _calibrate_gate(target_us: float = 25.0) -> int:
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

    def start(self) -> torch.cuda.Event:
        # An eager step is CPU-bound: without a gate the start event retires before the kernel packet
        # has even been submitted and the pair measures launch latency, not the kernel.  A short
        # device-side spin (~20 us) lets the CPU queue start-event, kernel(s) and stop-event first.
        if self.gate_cycles == 0:
            self.gate_cycles = self._calibrate_gate()
        torch.cuda._sleep(self.gate_cycles)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def stop(self, name: str, start: torch.cuda.Event, nbytes: float, flops: float = 0.0):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self._pending.append((name, start, e, float(nbytes), float(flops)))

    def _drain(self):
        if not self._pending:
            return
        torch.cuda.synchronize()
        for name, s, e, b, f in self._pending:
            a = self._acc[name]
            a[0] += s.elapsed_time(e)
            a[1] += b
            a[2] += f
            a[3] += 1
        self._pending.clear()

    def summary(self) -> Dict[str, dict]:
        self._drain()
        out = {}
        for name, (ms, b, f, n) in self._acc.items():
            if n == 0:
                continue
            out[name] = {"launches": n, "avg_us": ms * 1e3 / n, "bytes_per_launch": b / n,
                         "flops_per_launch": f / n, "gbps": (b / (ms * 1e-3)) / 1e9 if ms > 0 else 0.0,
                         "tflops": (f / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0}
        return out

    def reset(self):
        self._drain()
        self._acc.clear()
