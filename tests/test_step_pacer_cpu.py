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
