"""The decode-step deadline on the host (semi_pd/step_pacer.py, semi_pd/share_board.py): the decode instance publishes when its
step in flight began, the prefill instance's layer hook passes while there is none or it is young, holds while it is overdue
until the stamp changes, and never longer than MAX_WAIT_MS.  No GPU: the bounded run-ahead (HIP events) is covered by
tests/test_gpu_cu_share.py.  The reference has no counterpart (static MPS percentages, semi_pd/utils.py:10-11)."""
import os

import pytest

from semi_pd_amd.semi_pd import step_pacer as SP
from semi_pd_amd.semi_pd.share_board import ShareBoard


class FakeTime:
    def __init__(self):
        self.ns = 1_000_000_000
        self.sleeps = 0
        self.on_sleep = None

    def clock(self):
        return self.ns

    def sleep(self, s):
        self.sleeps += 1
        self.ns += int(s * 1e9)
        if self.on_sleep:
            self.on_sleep(self)


@pytest.fixture
def boards(tmp_path):
    path = os.path.join(tmp_path, "board")
    d, p = ShareBoard(path, create=True), ShareBoard(path, create=True)   # the two instances' mappings of one file
    yield d, p
    d.close(), p.close()


def test_the_step_in_flight_travels_over_the_board(boards):
    d, p = boards
    assert p.step_in_flight() == (0, 0)
    d.publish_step(123456)
    assert p.step_in_flight() == (123456, 1)
    d.publish_step(0)                      # nothing in flight: the sequence number stays
    assert p.step_in_flight() == (0, 1)
    d.publish_step(222)
    assert p.step_in_flight() == (222, 2)


def test_the_hook_passes_holds_and_times_out(boards):
    d, p = boards
    t = FakeTime()
    pacer = SP.StepPacer(p, deadline_ms=8.0, device=None, clock=t.clock, sleep=t.sleep)
    pacer.before_layer(0)                                       # no step in flight
    assert pacer.stats()["gates"] == 1 and pacer.stats()["holds"] == 0 and t.sleeps == 0
    d.publish_step(t.ns - 3_000_000)                            # a step 3 ms old: young
    pacer.before_layer(1)
    assert pacer.stats()["holds"] == 0 and t.sleeps == 0
    # 9 ms old: overdue.  The decode instance ends the step 2 ms later and starts the next one
    d.publish_step(t.ns - 9_000_000)
    t0 = t.ns

    def end_after_2ms(ft):
        if ft.ns - t0 >= 2_000_000:
            d.publish_step(ft.ns)
            ft.on_sleep = None
    t.on_sleep = end_after_2ms
    pacer.before_layer(2)
    st = pacer.stats()
    assert st["holds"] == 1 and st["timeouts"] == 0 and 1.9 <= st["held_ms"] <= 2.2, st
    # the new step is young again: the next layer passes at once
    n = t.sleeps
    pacer.before_layer(3)
    assert t.sleeps == n and pacer.stats()["holds"] == 1
    # an overdue step whose owner died: the hold ends by itself
    d.publish_step(t.ns - 20_000_000)
    pacer.before_layer(4)
    st = pacer.stats()
    assert st["holds"] == 2 and st["timeouts"] == 1 and SP.MAX_WAIT_MS <= st["held_ms"] - 2.0 <= SP.MAX_WAIT_MS + 1.0, st
    # ... and ONE timeout is the price: the later layers of this and every following forward pass at once while that stamp
    # stands (32 layers x 50 ms per batch otherwise)
    n = t.sleeps
    for layer in range(5, 40):
        pacer.before_layer(layer % 32)
    assert t.sleeps == n and pacer.stats()["holds"] == 2 and pacer.stats()["timeouts"] == 1
    # a step that merely ENDS (no successor) releases a hold as well
    d.publish_step(t.ns - 9_000_000)
    t.on_sleep = lambda ft: d.publish_step(0)
    pacer.before_layer(5)
    assert pacer.stats()["holds"] == 3 and pacer.stats()["timeouts"] == 1
    pacer.reset_stats()
    assert pacer.stats()["gates"] == 0 and pacer.stats()["held_ms"] == 0


def test_a_deadline_needs_the_share_board():
    from semi_pd_amd.server_args import ServerArgs
    with pytest.raises(ValueError, match="dynamic"):
        ServerArgs(enable_semi_pd=True, cu_mask_mode="env", decode_step_deadline_ms=8.0)
    assert ServerArgs(enable_semi_pd=True, decode_step_deadline_ms=8.0).cu_mask_mode == "dynamic"


def test_the_deadline_follows_the_objective(boards):
    """slo_ms: the pacer sees the steps that overlap prefill work begin and end, counts the token gaps above the objective
    against all steps of a window and moves the deadline towards the point where 1 % of the gaps exceed it."""
    from semi_pd_amd.semi_pd.share_board import BUSY_DECODE
    d, p = boards
    t = FakeTime()
    pacer = SP.StepPacer(p, deadline_ms=9.0, device=None, clock=t.clock, sleep=t.sleep, slo_ms=12.0)
    d.store(BUSY_DECODE, 24)

    def run_window(long_every):
        """SLO_WINDOW steps; every `long_every`-th lasts 14 ms (above the objective), the others 5 ms; the hook looks at the
        board once per step (a young step: no hold)."""
        for k in range(SP.SLO_WINDOW + 1):
            dur = 14_000_000 if (long_every and k % long_every == 0) else 5_000_000
            d.publish_step(t.ns)
            pacer.before_layer(1)
            t.ns += dur
    # 1 step in 20 is long: 5 % of the gaps above the objective -> the deadline comes down, two steps at once
    run_window(20)
    st = pacer.stats()
    assert st["slo_adjustments"] == 1 and st["share_of_gaps_over_slo"] == pytest.approx(0.05, abs=0.01)
    assert st["deadline_ms"] == pytest.approx(9.0 - 2 * SP.SLO_STEP_MS)
    # no long steps: it creeps back up (half a step)
    run_window(0)
    assert pacer.stats()["deadline_ms"] == pytest.approx(9.0 - 1.5 * SP.SLO_STEP_MS)
    # 1 in 60: 1.7 % -> one step down
    run_window(60)
    assert pacer.stats()["deadline_ms"] == pytest.approx(max(9.0 - 2.5 * SP.SLO_STEP_MS, 9.0 - SP.MAX_TIGHTEN_MS))
    # the controller is bounded around the deadline it was started with: what it may trade is a bounded amount of TTFT
    for _ in range(80):
        run_window(0)
    assert pacer.stats()["deadline_ms"] == pytest.approx(9.0 + SP.MAX_RELAX_MS)
    for _ in range(80):
        run_window(2)
    st = pacer.stats()
    assert st["deadline_ms"] == pytest.approx(9.0 - SP.MAX_TIGHTEN_MS)
    assert st["deadline_range_ms"] == [pytest.approx(9.0 - SP.MAX_TIGHTEN_MS), pytest.approx(9.0 + SP.MAX_RELAX_MS)]
    # every adjustment is on record (bounded: the first one and the most recent ones), as (decode step number, deadline)
    traj = st["deadline_trajectory"]
    assert 2 <= len(traj) <= SP.TRAJECTORY_KEEP and traj[0][1] == pytest.approx(9.0 - 2 * SP.SLO_STEP_MS)
    assert traj[-1][1] == st["deadline_ms"] and all(a[0] < b[0] for a, b in zip(traj, traj[1:]))
    # a start at the edge of the absolute range stays inside it
    edge = SP.StepPacer(p, deadline_ms=SP.DEADLINE_RANGE_MS[0], device=None, clock=t.clock, sleep=t.sleep, slo_ms=12.0)
    assert edge.stats()["deadline_range_ms"][0] == SP.DEADLINE_RANGE_MS[0]
    # without an objective the deadline is what was given
    fixed = SP.StepPacer(p, deadline_ms=9.0, device=None, clock=t.clock, sleep=t.sleep)
    run = fixed.stats()["deadline_ms"]
    assert run == 9.0 and "slo_adjustments" not in fixed.stats()
    from semi_pd_amd.server_args import ServerArgs
    with pytest.raises(ValueError, match="starting deadline"):
        ServerArgs(enable_semi_pd=True, decode_step_deadline_ms=0.0, decode_tbt_slo_ms=12.0)
    # the defaults: the measured operating point in dynamic mode, nothing where there is no share board
    from semi_pd_amd import server_args as SA
    a = ServerArgs(enable_semi_pd=True)
    assert (a.decode_step_deadline_ms, a.decode_tbt_slo_ms) == (SA.DEFAULT_DECODE_STEP_DEADLINE_MS, SA.DEFAULT_DECODE_TBT_SLO_MS)
    b = ServerArgs(enable_semi_pd=True, cu_mask_mode="env")
    assert (b.decode_step_deadline_ms, b.decode_tbt_slo_ms) == (0.0, 0.0) and ServerArgs().decode_step_deadline_ms == 0.0


def test_nothing_is_asked_of_a_step_that_it_could_not_do_alone(boards):
    """STEP_FAST_NS (the decode instance's 10th-percentile step time) floors the deadline and the objective: a model whose
    step takes 8 ms alone is not held at 8.5 ms, and its 10 ms steps do not count against a 12 ms objective."""
    d, p = boards
    t = FakeTime()
    pacer = SP.StepPacer(p, deadline_ms=8.5, device=None, clock=t.clock, sleep=t.sleep, slo_ms=12.0)
    d.publish_fast_step(8_000_000)
    d.publish_step(t.ns - 10_000_000)          # 10 ms old: past the fixed deadline, inside 1.4 x 8 ms
    pacer.before_layer(0)
    assert pacer.stats()["holds"] == 0
    d.publish_step(t.ns - 12_000_000)          # past 11.2 ms: held
    t.on_sleep = lambda ft: d.publish_step(ft.ns)
    pacer.before_layer(1)
    assert pacer.stats()["holds"] == 1
    d.publish_fast_step(4_400_000)             # a Llama-3-8B-sized step: the fixed deadline is the one in force
    d.publish_step(t.ns - 9_000_000)
    t.on_sleep = lambda ft: d.publish_step(ft.ns)
    pacer.before_layer(2)
    assert pacer.stats()["holds"] == 2


def test_a_silent_decode_instance_holds_nobody(boards):
    """A decode instance that died with a step published: its heartbeat (BEAT_DECODE, refreshed with every step) goes stale and
    the hooks stop holding for its stamp -- share_board.peer_busy's rule for BUSY_DECODE, applied to STEP_START_NS."""
    from semi_pd_amd.semi_pd.share_board import BEAT_DECODE
    d, p = boards
    t = FakeTime()
    pacer = SP.StepPacer(p, deadline_ms=8.0, device=None, clock=t.clock, sleep=t.sleep)
    d.store(BEAT_DECODE, t.ns - 3_000_000_000)          # last heard of 3 s ago (stale_s = 2)
    d.publish_step(t.ns - 20_000_000)
    pacer.before_layer(0)
    assert pacer.stats()["holds"] == 0 and t.sleeps == 0
    d.store(BEAT_DECODE, t.ns - 1_000_000)              # alive: the same stamp holds
    t.on_sleep = lambda ft: d.publish_step(ft.ns)
    pacer.before_layer(1)
    assert pacer.stats()["holds"] == 1 and pacer.stats()["timeouts"] == 0


def test_the_pacer_is_off_under_tensor_parallelism():
    """every rank would hold on its own board with no agreement while the others have launched the layer's
    peer-memory all-reduce.  Until a rank-0 decision is broadcast (as CuShare.decide does for the share), tp_size > 1 runs
    without the deadline; asking for one is refused."""
    from semi_pd_amd.server_args import ServerArgs
    a = ServerArgs(enable_semi_pd=True, tp_size=2)
    assert (a.decode_step_deadline_ms, a.decode_tbt_slo_ms) == (0.0, 0.0)
    with pytest.raises(ValueError, match="tp_size"):
        ServerArgs(enable_semi_pd=True, tp_size=2, decode_step_deadline_ms=8.5)


def test_the_decode_scheduler_publishes_its_fast_step(boards, monkeypatch):
    """managers/semi_pd_decode_scheduler.py: _publish_step keeps the step stamps and, every 64 steps, the 10th percentile of
    the step times on the board."""
    import time as _time
    from types import SimpleNamespace
    from semi_pd_amd.managers.semi_pd_decode_scheduler import SemiPDDecodeScheduler
    d, p = boards
    now = [5_000_000_000]
    monkeypatch.setattr(_time, "monotonic_ns", lambda: now[0])
    sched = SemiPDDecodeScheduler.__new__(SemiPDDecodeScheduler)
    sched.model_runner = SimpleNamespace(cu_share=SimpleNamespace(board=d))
    for k in range(65):
        sched._publish_step(True)
        now[0] += 4_000_000 if k % 2 else 9_000_000     # half the steps 9 ms, half 4 ms
    assert p.fast_step_ns() == 4_000_000 and p.step_in_flight()[1] == 65
    sched._publish_step(False)
    assert p.step_in_flight()[0] == 0


def test_which_steps_are_sampled_depends_on_the_step_count_alone():
    """model_executor/kernel_timing.py: a sampled step runs eagerly on the raw batch, the others replay a graph of a padded
    batch; under tensor parallelism all ranks must agree on which is which (the peer-memory collectives of the two paths have
    different block counts).  A rank whose samples nobody collects (a full pending list) samples the SAME steps as rank 0 --
    it only stops recording."""
    from semi_pd_amd.model_executor.kernel_timing import KernelTiming
    drained, full = KernelTiming(sample_every=4, max_pending=8), KernelTiming(sample_every=4, max_pending=8)
    full._pending = [None] * 8
    seq_a = [drained.begin_step() for _ in range(40)]
    seq_b = [full.begin_step() for _ in range(40)]
    assert seq_a == seq_b and sum(seq_a) == 10
    full.stop("x", None, 1.0)          # dropped: no event is created for it (none could be on this CPU)
    assert len(full._pending) == 8
