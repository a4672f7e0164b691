"""Decode-step deadline gate (csrc/step_clock.hip): the decode instance stamps a shared device slot at the start of every
step and clears it at the end; the prefill instance launches a one-wave gate kernel between decoder layers that holds its
compute stream while a decode step is older than the deadline.

The reference has no counterpart: its compute split is two static MPS percentages (semi_pd/utils.py:10-11,
entrypoints/engine.py:588-634).  Here the prefill share can be the large one (224 of 256 CUs: the time to the first token)
because the decode tail it would cause is cut where it forms (ServerArgs.decode_step_deadline_ms).

One slot per GPU: allocated by the decode process of that GPU (uncached device memory, like the all-reduce flags), exported
next to the KV pool's handles (IPCInfo.kvcache_info["step_clock"]), opened by the prefill process of the same GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from semi_pd_amd import _lib

SLOT_BYTES = 256
MAX_WAIT_MS = 50.0      # a gate never holds longer than this: a decode instance that died must not stall its neighbour


class StepClock:
    def __init__(self, device: torch.device, slot_ptr: int, owner: bool, peer_base: Optional[int] = None):
        self.device = torch.device(device)
        self.slot = int(slot_ptr)
        self.owner = owner
        self._peer_base = peer_base
        ticks = C.c_uint64()
        _lib.check(_lib.load().semipd_step_clock_ticks_per_ms(self.device.index or 0, C.addressof(ticks)),
                   "step_clock_ticks_per_ms")
        self.ticks_per_ms = int(ticks.value)
        self.deadline_ticks = 0
        self._stats = None

    # ---- decode instance -------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, device: torch.device) -> "StepClock":
        lib = _lib.load()
        with torch.cuda.device(device):
            p = C.c_void_p()
            _lib.check(lib.semipd_ar_alloc_shared(SLOT_BYTES, C.addressof(p)), "ar_alloc_shared")
        return cls(device, p.value, owner=True)

    def export(self) -> dict:
        handle = (C.c_uint8 * 64)()
        offset = C.c_uint64()
        _lib.check(_lib.load().semipd_ipc_get_handle(self.slot, C.addressof(handle), C.addressof(offset)), "ipc_get_handle")
        return {"handle": list(bytes(handle)), "offset": int(offset.value)}

    def peek(self):
        """(stamp of the step in flight or 0, steps begun): a synchronous device-to-host copy, for tests and diagnostics."""
        buf = (C.c_uint64 * 2)()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(buf, C.c_void_p(self.slot), 16, 2)    # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError(f"hipMemcpy of the step-clock slot failed ({rc})")
        return int(buf[0]), int(buf[1])

    def mark(self, begin: bool) -> None:
        """One-wave kernel on the current stream (captured into the decode graph): stamp / clear the slot."""
        _lib.check(_lib.load().semipd_step_clock_mark(self.slot, 1 if begin else 0, _lib.current_stream(self.device)),
                   "step_clock_mark")

    # ---- prefill instance ------------------------------------------------------------------------------------------
    @classmethod
    def open(cls, device: torch.device, exported: dict) -> "StepClock":
        lib = _lib.load()
        hb = (C.c_uint8 * 64).from_buffer_copy(bytes(exported["handle"]))
        base = C.c_void_p()
        dev = torch.device(device)
        with torch.cuda.device(dev):
            _lib.check(lib.semipd_ipc_open(C.addressof(hb), dev.index or 0, C.addressof(base)), "ipc_open")
        return cls(dev, base.value + int(exported["offset"]), owner=False, peer_base=base.value)

    def set_deadline_ms(self, ms: float) -> None:
        self.deadline_ticks = max(0, int(ms * self.ticks_per_ms))
        if self._stats is None:
            self._stats = torch.zeros(4, dtype=torch.int64, device=self.device)

    def gate(self) -> None:
        """One-wave kernel on the current stream: holds it while a decode step is older than the deadline."""
        if self.deadline_ticks <= 0:
            return
        _lib.check(_lib.load().semipd_step_clock_gate(self.slot, self.deadline_ticks, int(MAX_WAIT_MS * self.ticks_per_ms),
                                                      _lib.ptr(self._stats), _lib.current_stream(self.device)),
                   "step_clock_gate")

    def stats(self) -> dict:
        """{gates, holds, held_ms, timeouts} so far (synchronises the device: statistics requests only)."""
        if self._stats is None:
            return {}
        g, h, t, to = (int(v) for v in self._stats.cpu().tolist())
        return {"gates": g, "holds": h, "held_ms": round(t / self.ticks_per_ms, 3), "timeouts": to}

    def reset_stats(self) -> None:
        if self._stats is not None:
            self._stats.zero_()

    def close(self) -> None:
        lib = _lib.load()
        if self.owner and self.slot:
            lib.semipd_ar_free_shared(self.slot)
        elif self._peer_base:
            lib.semipd_ipc_close(self._peer_base)
        self.slot = 0
