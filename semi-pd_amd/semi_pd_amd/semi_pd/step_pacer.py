"""Decode-step deadline: the prefill instance paces its own launches and stops launching while a decode step is overdue.

What it stands in for.  The reference's compute split is two static MPS percentages (semi_pd/utils.py:10-11, entrypoints/
engine.py:588-634: prefill 80 %, decode 100 %): the decode tail under a prefill batch is whatever the non-shared percentage
delivers.  On MI355X that is a hard trade: next to a 224-CU prefill share a decode step streams its 19 GB
through the 32 CUs the share leaves free and takes 11-15 ms instead of 4.5 -- TTFT p50 29 ms, TBT p99 15.6 ms -- while the
192-CU share that keeps the tail at 11.6 ms costs 11 ms of TTFT.  A deadline cuts the tail where it forms instead of paying
for it with CUs all the time: when the decode step in flight is older than `deadline_ms`, the prefill instance stops
launching at its next layer boundary, the GPU drains its queue (at most `RUN_AHEAD` layers), the step finishes on the whole
chip, and launching resumes with the next step's stamp.

How.  Entirely on the host (a GPU-side hold -- a sleeping wave in the prefill stream -- was built first and made the decode
instance three times slower during the very holds that should have freed it; profiles/r05_deadline_gate_gpu_side_negative.txt):
  * the decode instance's host publishes when its step in flight began (semi_pd/share_board.py: STEP_START_NS / STEP_SEQ;
    managers/semi_pd_decode_scheduler.py) -- it knows: a step begins when its predecessor's result event fires, or at its
    launch when nothing ran;
  * the prefill instance's forward has a pre-hook in front of every decoder layer (ModelRunner.init_step_pacer):
      1. bounded run-ahead: an event is recorded per hook and the hook waits for the event of RUN_AHEAD hooks ago, so the
         GPU's queue never holds more than RUN_AHEAD layers (a 1 k-token layer is ~0.8 ms of GPU work and ~0.16 ms of host
         launches: the GPU never waits for the host, and a hold takes effect within RUN_AHEAD layers);
      2. the deadline: if the published step is older than the deadline the hook sleeps until the stamp changes (the step
         ended: the next one began, or none is in flight) or MAX_WAIT_MS have passed (a decode instance that died must not
         stall its neighbour).
A hold is therefore an EMPTY prefill queue: the state between two prefill batches, in which the decode instance is known to
run at full speed.  The late-binding loop of the prefill scheduler keeps working: the forward now returns RUN_AHEAD layers
before the GPU finishes the batch instead of ~20 ms before, which is still ahead of the moment the next batch is proposed.
"""
from __future__ import annotations

import collections
import time

import torch

import os

RUN_AHEAD = int(os.environ.get("SEMIPD_PACER_RUN_AHEAD", "2"))   # decoder layers the host may be ahead of the GPU
MAX_WAIT_MS = 50.0      # a hold never lasts longer than this


class StepPacer:
    def __init__(self, board, deadline_ms: float, device, run_ahead: int = RUN_AHEAD, clock=time.monotonic_ns, sleep=time.sleep):
        self.board = board
        self.deadline_ns = int(deadline_ms * 1e6)
        self.device = device
        self.run_ahead = int(run_ahead)
        self._clock, self._sleep = clock, sleep
        # the scheduler switches holds off for a batch it runs under the backlog rule (an overloaded GPU: throughput first --
        # every step of an overloaded decode instance is overdue, and holding for each of them costs 15 % of the capacity)
        self.hold_enabled = True
        self._ring = collections.deque()
        self._free = []
        self._stats = {"gates": 0, "holds": 0, "held_ms": 0.0, "timeouts": 0, "run_ahead_waits_ms": 0.0}

    # ---- the hook ----------------------------------------------------------------------------------------------------
    def before_layer(self, index: int) -> None:
        st = self._stats
        st["gates"] += 1
        if index == 0:                       # a new forward: nothing of the previous one bounds this one
            self._free.extend(self._ring)
            self._ring.clear()
        self._bound_run_ahead()
        self._hold_while_overdue()

    def _bound_run_ahead(self) -> None:
        if self.run_ahead <= 0 or self.device is None or torch.device(self.device).type != "cuda":
            return
        ev = self._free.pop() if self._free else torch.cuda.Event()
        ev.record()                          # everything launched so far = the layers before this hook
        self._ring.append(ev)
        if len(self._ring) > self.run_ahead:
            old = self._ring.popleft()
            if not old.query():
                t0 = time.perf_counter()
                old.synchronize()
                self._stats["run_ahead_waits_ms"] += (time.perf_counter() - t0) * 1e3
            self._free.append(old)

    def _hold_while_overdue(self) -> None:
        if self.board is None or self.deadline_ns <= 0 or not self.hold_enabled:
            return
        start, seq = self.board.step_in_flight()
        if not start:
            return
        t0 = self._clock()
        if t0 - start < self.deadline_ns:
            return
        st = self._stats
        st["holds"] += 1
        now = t0
        while True:
            s2, q2 = self.board.step_in_flight()
            if s2 != start or q2 != seq:
                break
            self._sleep(50e-6)
            now = self._clock()
            if (now - t0) > MAX_WAIT_MS * 1e6:
                st["timeouts"] += 1
                break
        st["held_ms"] += (now - t0) / 1e6

    # ---- statistics --------------------------------------------------------------------------------------------------
    def stats(self) -> dict:
        out = dict(self._stats)
        out["held_ms"] = round(out["held_ms"], 3)
        out["run_ahead_waits_ms"] = round(out["run_ahead_waits_ms"], 3)
        return out

    def reset_stats(self) -> None:
        for k in self._stats:
            self._stats[k] = 0 if isinstance(self._stats[k], int) else 0.0
