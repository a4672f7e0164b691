"""Decode-step deadline: the prefill instance paces its own launches and stops launching while a decode step is overdue.

What it stands in for.  The reference's compute split is two static MPS percentages (semi_pd/utils.py:10-11, entrypoints/
engine.py:588-634: prefill 80 %, decode 100 %): the decode tail under a prefill batch is whatever the non-shared percentage
delivers.  On MI355X that is a hard trade: next to a 224-CU prefill share a decode step streams its 19 GB
through the 32 CUs the share leaves free and takes 11-15 ms instead of 4.5 -- TTFT p50 29 ms, TBT p99 15.6 ms -- while the
192-CU share that keeps the tail at 11.6 ms costs 11 ms of TTFT.  A deadline cuts the tail where it forms instead of paying
for it with CUs all the time: when the decode step in flight is older than `deadline_ms`, the prefill instance stops
launching at its next layer boundary, the GPU drains its queue (at most `RUN_AHEAD` layers: one), the step finishes on the whole
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
Optionally the deadline follows a service-level objective (`slo_ms`, ServerArgs.decode_tbt_slo_ms): the hooks see every
decode step that overlaps prefill work begin (STEP_SEQ / STEP_START_NS), so the pacer knows how long each of them took and
how many steps there were in all; every SLO_WINDOW steps it compares the share of token gaps above the objective (steps
weighted by their batch size, BUSY_DECODE) with 1 % and moves the deadline by SLO_STEP_MS (twice that when far off) towards the point where the 99th
percentile sits on the objective -- the latest deadline, i.e. the fewest and shortest holds, that still keeps the tail.

A hold is therefore an EMPTY prefill queue: the state between two prefill batches, in which the decode instance is known to
run at full speed.  The late-binding loop of the prefill scheduler keeps working: the forward now returns RUN_AHEAD layers
before the GPU finishes the batch instead of ~20 ms before, which is still ahead of the moment the next batch is proposed.
"""
from __future__ import annotations

import collections
import time

import torch

import os

RUN_AHEAD = int(os.environ.get("SEMIPD_PACER_RUN_AHEAD", "1"))   # decoder layers the host may be ahead of the GPU
MAX_WAIT_MS = 50.0      # a hold never lasts longer than this
SLO_WINDOW = 256        # decode steps per adjustment of the deadline (about 1.5 s at 6 ms per step)
SLO_STEP_MS = 0.25
SLO_MARGIN_MS = 0.3     # the client's token gap is the step plus host work of the decode instance
DEADLINE_RANGE_MS = (5.0, 16.0)
# Nothing can be asked of a decode step that it could not do alone: the deadline in force is at least FLOOR_DEADLINE times, the
# objective at least FLOOR_SLO times the decode instance's fast step (STEP_FAST_NS: Llama-3-8B ~4.4 ms, DeepSeek-V2-Lite
# ~8 ms -- with a fixed 8.5 ms every one of its steps would be "overdue" and the prefill instance held for nothing)
FLOOR_DEADLINE = 1.4
FLOOR_SLO = 1.8


class StepPacer:
    def __init__(self, board, deadline_ms: float, device, run_ahead: int = RUN_AHEAD, clock=time.monotonic_ns, sleep=time.sleep,
                 slo_ms: float = 0.0):
        self.board = board
        self.deadline_ns = int(deadline_ms * 1e6)
        self.slo_ns = int(max(0.0, slo_ms - SLO_MARGIN_MS) * 1e6) if slo_ms and slo_ms > 0 else 0
        self._seen = None                 # (start, seq, batch) of the step last seen in flight
        self._win = [0, 0.0, 0.0]         # this window: first seq, token gaps in all, token gaps above the objective
        self.device = device
        self.run_ahead = int(run_ahead)
        self._clock, self._sleep = clock, sleep
        # the scheduler switches holds off for a batch it runs under the backlog rule (an overloaded GPU: throughput first --
        # every step of an overloaded decode instance is overdue, and holding for each of them costs 15 % of the capacity)
        self.hold_enabled = True
        # called while the hook waits (for the GPU or for an overdue step): the prefill scheduler sends the first tokens of the
        # batch that ran before this one the moment its event fires, instead of at the next layer boundary (~0.8 ms later)
        self.while_waiting = None
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
                if self.while_waiting is None:
                    old.synchronize()
                else:
                    while not old.query():
                        self.while_waiting()
                        time.sleep(20e-6)
                self._stats["run_ahead_waits_ms"] += (time.perf_counter() - t0) * 1e3
            self._free.append(old)

    def _hold_while_overdue(self) -> None:
        if self.board is None or self.deadline_ns <= 0 or not self.hold_enabled:
            return
        start, seq = self.board.step_in_flight()
        if self.slo_ns:
            self._observe(start, seq)
        if not start:
            return
        t0 = self._clock()
        fast = self.board.fast_step_ns()
        if t0 - start < max(self.deadline_ns, int(FLOOR_DEADLINE * fast)):
            return
        st = self._stats
        st["holds"] += 1
        now = t0
        while True:
            s2, q2 = self.board.step_in_flight()
            if s2 != start or q2 != seq:
                break
            if self.while_waiting is not None:
                self.while_waiting()
            self._sleep(50e-6)
            now = self._clock()
            if (now - t0) > MAX_WAIT_MS * 1e6:
                st["timeouts"] += 1
                break
        st["held_ms"] += (now - t0) / 1e6

    # ---- the deadline follows the objective ------------------------------------------------------------------------
    def _observe(self, start: int, seq: int) -> None:
        """Called with what the board shows at a hook.  A step whose successor's stamp is seen has a known duration."""
        from semi_pd_amd.semi_pd.share_board import BUSY_DECODE
        prev = self._seen
        if prev is not None and seq != prev[1]:
            if start and seq == prev[1] + 1 and prev[0]:
                dur = start - prev[0]                       # step prev ended where its successor began
                if dur > max(self.slo_ns, int(FLOOR_SLO * self.board.fast_step_ns())):
                    self._win[2] += prev[2]
            self._seen = None
        if start and self._seen is None:
            self._seen = (start, seq, max(1, self.board.load(BUSY_DECODE)))
        if self._win[0] == 0:
            self._win[0] = seq
        steps = seq - self._win[0]
        if steps >= SLO_WINDOW:
            # token gaps in the window: steps x the batch size seen last (steps nobody watched ran without prefill work
            # next to them: short ones)
            batch = self._seen[2] if self._seen else max(1, self.board.load(BUSY_DECODE))
            total = max(1.0, steps * float(batch))
            over = self._win[2] / total
            lo, hi = DEADLINE_RANGE_MS
            # down by one step while more than 1 % of the gaps exceed the objective (two steps from 2.5 %), up by half a step
            d = self.deadline_ns / 1e6 + (-2 * SLO_STEP_MS if over > 0.025 else -SLO_STEP_MS if over > 0.01 else SLO_STEP_MS / 2)
            self.deadline_ns = int(min(hi, max(lo, d)) * 1e6)
            self._stats["slo_adjustments"] = self._stats.get("slo_adjustments", 0) + 1
            self._stats["share_of_gaps_over_slo"] = round(over, 4)
            self._win = [seq, 0.0, 0.0]

    # ---- statistics --------------------------------------------------------------------------------------------------
    def stats(self) -> dict:
        out = dict(self._stats)
        out["held_ms"] = round(out["held_ms"], 3)
        out["run_ahead_waits_ms"] = round(out["run_ahead_waits_ms"], 3)
        out["deadline_ms"] = round(self.deadline_ns / 1e6, 3)
        return out

    def reset_stats(self) -> None:
        for k in list(self._stats):
            self._stats[k] = 0 if isinstance(self._stats[k], int) else 0.0
