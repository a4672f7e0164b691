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
The controller is BOUNDED around the deadline it was started with (MAX_TIGHTEN_MS below, MAX_RELAX_MS above): on the frontier of
DESIGN.md 4.5 a millisecond of TBT tail costs ~4 ms of TTFT p50, and on a slow box an unbounded controller bought 0.3 ms of
tail with 1.25 ms of deadline and ~5 ms of TTFT (BENCH_r05: 8.5 -> 7.25 ms in 100 adjustments); bounded, what it can trade
is at most ~2 ms of TTFT.  Every adjustment is kept (stats()["deadline_trajectory"]) so that a run shows where it went.

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
# ... and the objective may move the deadline this far from the one the pacer was started with, no further
MAX_TIGHTEN_MS = 0.5
MAX_RELAX_MS = 2.0
TRAJECTORY_KEEP = 64    # adjustments remembered for the statistics (first + the most recent ones)
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
        lo, hi = DEADLINE_RANGE_MS
        # the controller's range: the absolute one, cut to [start - MAX_TIGHTEN_MS, start + MAX_RELAX_MS] (a start outside the
        # absolute range stays where it was put: the range then is that single point's side of it)
        self._range_ms = (min(deadline_ms, max(lo, deadline_ms - MAX_TIGHTEN_MS)), max(deadline_ms, min(hi, deadline_ms + MAX_RELAX_MS)))
        self._trajectory = []             # (decode step number, deadline in ms) after every adjustment
        self._gave_up = None              # (start, seq) of a step a hold timed out on: its owner is taken for dead
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
        # time_layers (bench.py's accounting of a prefill batch; ModelRunner.init_step_pacer with kernel timing collected): the
        # events of the run-ahead bound carry timestamps, and the GPU time between two consecutive hooks -- one decoder layer's
        # kernels and the launch gaps inside it, plus the idle time of a hold at the first of the two -- is summed separately for
        # intervals with and without a hold
        self.time_layers = False
        self._last_done = None            # (event, hook index, held at that hook) of the newest completed hook
        self._held_at = set()             # hook indices of this forward at which a hold happened
        self._layers = {"intervals": 0, "ms": 0.0, "held_intervals": 0, "held_interval_ms": 0.0}

    # ---- the hook ----------------------------------------------------------------------------------------------------
    def before_layer(self, index: int) -> None:
        st = self._stats
        st["gates"] += 1
        if index == 0:                       # a new forward: nothing of the previous one bounds this one
            self._free.extend(e for e, _ in self._ring)
            self._ring.clear()
            if self._last_done is not None:
                self._free.append(self._last_done[0])
                self._last_done = None
            self._held_at.clear()
        self._bound_run_ahead(index)
        holds = self._stats["holds"]
        self._hold_while_overdue()
        if self.time_layers and self._stats["holds"] != holds:
            self._held_at.add(index)

    def _bound_run_ahead(self, index: int = 0) -> None:
        if self.run_ahead <= 0 or self.device is None or torch.device(self.device).type != "cuda":
            return
        ev = self._free.pop() if self._free else torch.cuda.Event(enable_timing=self.time_layers)
        ev.record()                          # everything launched so far = the layers before this hook
        self._ring.append((ev, index))
        if len(self._ring) > self.run_ahead:
            old, old_index = self._ring.popleft()
            if not old.query():
                t0 = time.perf_counter()
                if self.while_waiting is None:
                    old.synchronize()
                else:
                    while not old.query():
                        self.while_waiting()
                        time.sleep(20e-6)
                self._stats["run_ahead_waits_ms"] += (time.perf_counter() - t0) * 1e3
            if not self.time_layers:
                self._free.append(old)
                return
            prev = self._last_done
            if prev is not None and prev[1] + 1 == old_index:
                # hook prev -> hook old: layer `prev[1]` on the GPU (+ the idle time of a hold at hook prev[1])
                dt = prev[0].elapsed_time(old)
                acc = self._layers
                if prev[1] in self._held_at:
                    acc["held_intervals"] += 1
                    acc["held_interval_ms"] += dt
                else:
                    acc["intervals"] += 1
                    acc["ms"] += dt
            if prev is not None:
                self._free.append(prev[0])
            self._last_done = (old, old_index)

    def _hold_while_overdue(self) -> None:
        if self.board is None or self.deadline_ns <= 0 or not self.hold_enabled:
            return
        start, seq = self.board.step_in_flight()
        if self.slo_ns:
            self._observe(start, seq)
        if not start:
            return
        if self._gave_up is not None:
            # a hold on this very stamp ran into MAX_WAIT_MS: the decode instance died or hangs with a step published.  One
            # timeout is the price; every later layer passes until the stamp changes (32 layers x 50 ms per batch otherwise)
            if self._gave_up == (start, seq):
                return
            self._gave_up = None
        t0 = self._clock()
        if self._decode_silent(t0):
            return
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
                self._gave_up = (start, seq)
                break
        st["held_ms"] += (now - t0) / 1e6

    def _decode_silent(self, now_ns: int) -> bool:
        """The decode instance's heartbeat (BEAT_DECODE, written with every step it publishes) is older than the board's
        staleness bound: whatever STEP_START_NS says, no step of a live instance is in flight (share_board.peer_busy applies
        the same rule to BUSY_DECODE)."""
        from semi_pd_amd.semi_pd.share_board import BEAT_DECODE
        stale = getattr(self.board, "stale_ns", 0)
        if not stale:
            return False
        beat = self.board.load(BEAT_DECODE)
        return bool(beat) and now_ns - beat > stale

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
            lo, hi = self._range_ms
            # down by one step while more than 1 % of the gaps exceed the objective (two steps from 2.5 %), up by half a step
            d = self.deadline_ns / 1e6 + (-2 * SLO_STEP_MS if over > 0.025 else -SLO_STEP_MS if over > 0.01 else SLO_STEP_MS / 2)
            self.deadline_ns = int(min(hi, max(lo, d)) * 1e6)
            self._stats["slo_adjustments"] = self._stats.get("slo_adjustments", 0) + 1
            self._stats["share_of_gaps_over_slo"] = round(over, 4)
            self._trajectory.append((int(seq), round(self.deadline_ns / 1e6, 3)))
            if len(self._trajectory) > TRAJECTORY_KEEP:
                del self._trajectory[1]      # keep the first adjustment and the most recent ones
            self._win = [seq, 0.0, 0.0]

    # ---- statistics --------------------------------------------------------------------------------------------------
    def stats(self) -> dict:
        out = dict(self._stats)
        out["held_ms"] = round(out["held_ms"], 3)
        out["run_ahead_waits_ms"] = round(out["run_ahead_waits_ms"], 3)
        out["deadline_ms"] = round(self.deadline_ns / 1e6, 3)
        if self.slo_ns:
            out["deadline_range_ms"] = [round(self._range_ms[0], 3), round(self._range_ms[1], 3)]
            out["deadline_trajectory"] = [list(x) for x in self._trajectory]
        if self.time_layers and self._layers["intervals"]:
            a = self._layers
            out["layer_ms_without_hold"] = round(a["ms"] / a["intervals"], 4)
            out["layer_intervals_timed"] = a["intervals"]
            if a["held_intervals"]:
                out["layer_ms_with_hold"] = round(a["held_interval_ms"] / a["held_intervals"], 4)
                out["layer_intervals_with_hold"] = a["held_intervals"]
        return out

    def reset_stats(self) -> None:
        for k in list(self._stats):
            self._stats[k] = 0 if isinstance(self._stats[k], int) else 0.0
        self._trajectory = []
        self._layers = {"intervals": 0, "ms": 0.0, "held_intervals": 0, "held_interval_ms": 0.0}
        self._layers = {"intervals": 0, "ms": 0.0, "held_intervals": 0, "held_interval_ms": 0.0}
