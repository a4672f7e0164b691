"""Share board of one GPU's prefill / decode pair (include/semipd.h: semipd_share_board_*).

The reference gives its instances OVERLAPPING shares -- prefill 80 % of the SMs, decode 100 % (semi_pd/utils.py:10-11,
entrypoints/engine.py:588-593, 632-634) -- and lets MPS time-share what overlaps, so its decode instance is never confined to a
slice.  CU masks are hard partitions: a masked decode instance keeps to its slice even while the prefill instance has
nothing to do.  Here each instance publishes how much work it has in flight on one page of shared host memory, and picks --
per decode step / per prefill batch -- between its own share and the whole chip (model_executor/cu_share.py).

Slots (int64 each, one writer per slot):
    BUSY_PREFILL   prefill batches launched and not yet finished
    BUSY_DECODE    requests in the decode instance's running batch (0 = no step in flight)
    BEAT_*         time.monotonic_ns() of the writer's last update (a reader treats a silent peer as idle after `stale_s`)
    TAKEN_*        counters for the statistics: steps / batches the instance ran on the whole chip
    STEP_START_NS  time.monotonic_ns() at which the decode step in flight began on the GPU (0 = none), STEP_SEQ its number:
                   what the prefill instance's step pacer reads (semi_pd/step_pacer.py)
    STEP_FAST_NS   how long a decode step takes when nothing is in its way: the 10th percentile of the decode instance's
                   recent step times (a deadline below a small multiple of it could never be met: the pacer's floor)
"""
from __future__ import annotations

import ctypes as C
import time

from semi_pd_amd import _lib
from semi_pd_amd.semi_pd.utils import InstanceRole

BUSY_PREFILL, BUSY_DECODE, BEAT_PREFILL, BEAT_DECODE, TAKEN_PREFILL, TAKEN_DECODE, STEP_START_NS, STEP_SEQ, STEP_FAST_NS = range(9)


class ShareBoard:
    def __init__(self, path: str, create: bool = False, stale_s: float = 2.0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.semipd_share_board_open(path.encode(), 1 if create else 0, C.addressof(h)), "share_board_open")
        self._h = h.value
        self.path = path
        self.stale_ns = int(stale_s * 1e9)

    def close(self):
        if self._h:
            self._lib.semipd_share_board_close(self._h)
            self._h = None

    def store(self, slot: int, value: int) -> None:
        _lib.check(self._lib.semipd_share_board_store(self._h, slot, int(value)), "share_board_store")

    def add(self, slot: int, delta: int) -> int:
        out = C.c_int64()
        _lib.check(self._lib.semipd_share_board_add(self._h, slot, int(delta), C.addressof(out)), "share_board_add")
        return int(out.value)

    def load(self, slot: int) -> int:
        out = C.c_int64()
        _lib.check(self._lib.semipd_share_board_load(self._h, slot, C.addressof(out)), "share_board_load")
        return int(out.value)

    # ---- the two instances' view -------------------------------------------------------------------------------
    @staticmethod
    def _slots(role: InstanceRole):
        return (BUSY_PREFILL, BEAT_PREFILL) if role == InstanceRole.PREFILL else (BUSY_DECODE, BEAT_DECODE)

    def publish(self, role: InstanceRole, busy: int) -> None:
        """`role` has `busy` units of work in flight right now (0 = idle)."""
        b, t = self._slots(role)
        self.store(t, time.monotonic_ns())
        self.store(b, busy)

    # ---- the decode step in flight (written by the decode instance's host, read by the prefill instance's pacer) ----
    def publish_step(self, started_ns: int) -> None:
        """A decode step began on the GPU at `started_ns` (time.monotonic_ns(); 0 = no step in flight)."""
        if started_ns:
            self.add(STEP_SEQ, 1)
        self.store(STEP_START_NS, int(started_ns))

    def publish_fast_step(self, ns: int) -> None:
        self.store(STEP_FAST_NS, int(ns))

    def fast_step_ns(self) -> int:
        return self.load(STEP_FAST_NS)

    def step_in_flight(self):
        """(start in monotonic ns or 0, sequence number) of the decode step in flight."""
        return self.load(STEP_START_NS), self.load(STEP_SEQ)

    def peer_busy(self, role: InstanceRole) -> int:
        """Work the OTHER instance has in flight (0 = idle, or silent for longer than stale_s: a peer that died while busy
        must not confine this instance for ever)."""
        peer = InstanceRole.DECODE if role == InstanceRole.PREFILL else InstanceRole.PREFILL
        b, t = self._slots(peer)
        busy = self.load(b)
        if busy and time.monotonic_ns() - self.load(t) > self.stale_ns:
            return 0
        return busy
