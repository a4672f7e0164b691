"""The share board (include/semipd.h: semipd_share_board_*; semi_pd/share_board.py): one page of host memory through which
the two instances of a GPU tell each other whether they have work in flight.  No reference counterpart: it replaces what
MPS does with the reference's overlapping percentages (semi_pd/utils.py:10-11)."""
import multiprocessing as mp
import os
import time

import pytest

from semi_pd_amd.semi_pd.share_board import BUSY_DECODE, BUSY_PREFILL, TAKEN_DECODE, ShareBoard
from semi_pd_amd.semi_pd.utils import InstanceRole


def _child(path, n):
    b = ShareBoard(path)
    for i in range(n):
        b.add(TAKEN_DECODE, 1)
    b.publish(InstanceRole.PREFILL, 3)
    b.close()


def test_two_processes_meet_on_the_board(tmp_path):
    path = str(tmp_path / "share_board")
    a = ShareBoard(path, create=True)
    assert a.load(BUSY_PREFILL) == 0 and a.load(BUSY_DECODE) == 0      # a fresh board says "idle"
    assert a.peer_busy(InstanceRole.DECODE) == 0
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_child, args=(path, 1000)) for _ in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert a.load(TAKEN_DECODE) == 2000                                   # atomic adds from two processes
    assert a.peer_busy(InstanceRole.DECODE) == 3                          # the decode side sees the prefill side's work
    assert a.peer_busy(InstanceRole.PREFILL) == 0
    a.publish(InstanceRole.DECODE, 17)
    assert ShareBoard(path).peer_busy(InstanceRole.PREFILL) == 17
    a.close()


def test_a_silent_busy_peer_counts_as_idle(tmp_path):
    b = ShareBoard(str(tmp_path / "board"), create=True, stale_s=0.05)
    b.publish(InstanceRole.PREFILL, 1)
    assert b.peer_busy(InstanceRole.DECODE) == 1
    time.sleep(0.1)
    assert b.peer_busy(InstanceRole.DECODE) == 0      # it died while busy: nobody stays confined to a share for ever
    b.publish(InstanceRole.PREFILL, 1)
    assert b.peer_busy(InstanceRole.DECODE) == 1


def test_argument_errors(tmp_path):
    with pytest.raises(RuntimeError):
        ShareBoard(str(tmp_path / "missing"))           # not created: the opener must not invent a board
    b = ShareBoard(str(tmp_path / "board"), create=True)
    with pytest.raises(RuntimeError):
        b.store(64, 1)
    with pytest.raises(RuntimeError):
        b.load(-1)
    assert os.path.getsize(str(tmp_path / "board")) == 4096
