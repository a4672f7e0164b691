"""Work-conserving CU shares: the compute streams of ONE instance and the rule that picks between them.

Reference behaviour this stands in for: the MPS percentages of the two instances overlap (prefill 80 %, decode 100 %:
semi_pd/utils.py:10-11, set per process at entrypoints/engine.py:588-593, 632-634), so whichever instance is alone on the
GPU gets all of it.  HSA_CU_MASK is a hard, process-wide partition; `--cu-mask-mode dynamic` therefore leaves the process
unmasked and gives it TWO streams instead:

    "share"   hipExtStreamCreateWithCUMask over the instance's own CUs (prefill: the lowest P %, decode: the highest D %),
    "full"    the process's NULL stream: every CU of the device,

and the instance runs each unit of work (a decode step = one hipGraph launch, a prefill batch = one forward) on "full" while
the other instance has nothing in flight (semi_pd/share_board.py) and on "share" otherwise.  "full" is the NULL stream (see
__init__); with the reference's shares (prefill 80 %, decode 100 %) the decode instance therefore never leaves the NULL
stream, and the prefill instance moves between its masked stream and the NULL stream.  Kernels launched on a stream --
and the nodes of a hipGraph launched on it -- inherit the stream's CU mask (tools/cu_mask_check.py --streams verifies this
on the box).  Grids, K splits and split-KV counts are sized for the CUs of the stream they run on, so the decode instance
keeps one set of graphs per stream.

Ordering: the instance has ONE logical queue of work.  Whenever it moves from one stream to the other, the new stream first
waits for everything queued on the old one, so buffers allocated under one stream and read under the other are safe, and
the caching allocator (which reuses a block only on the stream it was allocated on) stays consistent.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Dict, Optional

import torch

from semi_pd_amd import _lib
from semi_pd_amd.semi_pd.utils import InstanceRole, cu_mask_words, cu_masked_stream

logger = logging.getLogger(__name__)

SHARE, FULL = "share", "full"


def agree_on_rank0(name: str, tp_rank: int, group) -> str:
    """Tensor parallelism: ONE stream choice per unit of work for all ranks of an instance -- rank 0's, sent with a
    broadcast on the gloo CPU group like the batch itself (managers/scheduler.py:645-659 broadcasts the scheduler's
    decisions the same way).  Each rank reads its own GPU's share board, and the boards need not agree at the instant
    they are read: ranks of one forward on different CU counts are harmless for the bits (the collectives do not care)
    but the slowest rank sets the time of every all-reduce."""
    from semi_pd_amd.distributed import broadcast_pyobj
    return broadcast_pyobj([name], tp_rank, group, src=0)[0]


class CuShare:
    def __init__(self, model_runner, role: InstanceRole, percent: int, board=None):
        self.mr = model_runner
        self.role = role
        self.board = board
        dev = model_runner.device
        idx = dev.index or 0
        from_top = role == InstanceRole.DECODE
        n = model_runner.num_cus
        self.cus: Dict[str, int] = {
            SHARE: sum(bin(w).count("1") for w in cu_mask_words(n, percent, from_top)) if percent < 100 else n,
            FULL: n}
        # The whole-chip stream is the process's NULL stream, on purpose: a decode instance whose graph launches go to a
        # CREATED stream is a much worse neighbour than one on the NULL stream -- same masks, same kernels: decode step
        # 6.4 -> 8.8 ms, TBT p99 12.4 -> 29.5 ms, prefill batch 29.8 -> 33.3 ms (profiles/r04_stream_kind_bisect.txt; it
        # also explains round 2's "a prioritised decode stream made both medians worse": that stream was a created one).
        # SEMIPD_DYN_FULL_ON_CREATED_STREAM=1 keeps the created stream for that A/B.
        if os.environ.get("SEMIPD_DYN_FULL_ON_CREATED_STREAM") == "1":
            raw = C.c_void_p()
            _lib.check(_lib.load().semipd_stream_create(idx, C.addressof(raw)), "stream_create")
            full = torch.cuda.ExternalStream(raw.value, device=dev)
        else:
            full = torch.cuda.default_stream(dev)
        self.streams: Dict[str, torch.cuda.Stream] = {
            SHARE: cu_masked_stream(idx, percent, from_top) if percent < 100 else full, FULL: full}
        self.active: Optional[str] = None
        self.taken = {SHARE: 0, FULL: 0}   # units of work run on each stream (statistics)
        # tensor parallelism: (tp_rank, gloo group) -- the choice is rank 0's for every rank (ModelRunner.init_cu_share)
        self.tp = None
        # the communication stream of the overlapped all-reduce (distributed.py) follows the compute stream: collectives
        # of an instance on its share run on a stream with the SAME CU mask (the reference's per-process MPS percentage
        # confines NCCL too, entrypoints/engine.py:591-593, 632-634), on an unmasked created stream on the whole chip
        self.comm_streams: Dict[str, Optional[torch.cuda.Stream]] = {
            SHARE: cu_masked_stream(idx, percent, from_top) if percent < 100 else None, FULL: None}
        self.activate(SHARE)

    # ---- which stream the next unit of work runs on ------------------------------------------------------------
    def choose(self) -> str:
        """The whole chip while the other instance is idle, the own share otherwise."""
        force = os.environ.get("SEMIPD_CU_SHARE_FORCE")      # experiments: "share" / "full" whatever the board says
        if force in (SHARE, FULL):
            return force
        if self.board is None or self.cus[SHARE] == self.cus[FULL]:
            return SHARE
        return FULL if self.board.peer_busy(self.role) == 0 else SHARE

    def decide(self, prefer: Optional[str] = None) -> str:
        """The stream of the next unit of work: `prefer` (the caller's own reason to take a stream, e.g. the prefill
        backlog) or choose(); under tensor parallelism rank 0's decision for every rank.  No GPU call in here."""
        name = prefer if prefer in (SHARE, FULL) else self.choose()
        if self.tp is not None and self.cus[SHARE] != self.cus[FULL]:
            name = agree_on_rank0(name, self.tp[0], self.tp[1])
        return name

    def activate(self, name: str) -> str:
        """Make `name` the stream every following launch of this thread goes to (torch's current stream; the C-ABI calls
        take it from there) and tell the kernels how many CUs it has."""
        new = self.streams[name]
        if name != self.active:
            if self.active is not None:
                new.wait_stream(self.streams[self.active])
            torch.cuda.set_stream(new)
            self.active = name
            self.mr.set_owned_cus(self.cus[name])
            from semi_pd_amd import distributed
            distributed.set_comm_stream(self.mr.device.index or 0, self.comm_streams[name],
                                        confined=self.cus[name] < self.cus[FULL])
        self.taken[name] += 1
        return name

    def step(self) -> str:
        return self.activate(self.decide())

    def close(self) -> None:
        """Orderly end of the instance: everything queued has run, torch is back on the NULL stream and the CU-masked stream
        is destroyed while the runtime is still up (a masked stream left to the process's exit handlers took the prefill
        process down with SIGSEGV inside __cxa_finalize -- and with it whatever a profiler attached to it had collected)."""
        dev = self.mr.device
        torch.cuda.synchronize(dev)
        full = torch.cuda.default_stream(dev)
        torch.cuda.set_stream(full)
        self.active = None
        from semi_pd_amd import distributed
        distributed.set_comm_stream(dev.index or 0, None, confined=False)
        for table in (self.streams, self.comm_streams):
            for name, st in list(table.items()):
                if isinstance(st, torch.cuda.ExternalStream) and st.cuda_stream != full.cuda_stream:
                    _lib.check(_lib.load().semipd_stream_destroy(C.c_void_p(st.cuda_stream)), "stream_destroy")
                table[name] = full if table is self.streams else None

    def publish(self, busy: int) -> None:
        if self.board is not None:
            self.board.publish(self.role, busy)
