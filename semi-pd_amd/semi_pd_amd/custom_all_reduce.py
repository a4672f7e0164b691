"""Tensor-parallel all-reduce over peer-mapped memory: the host side of csrc/all_reduce.hip.

Mirrors `CustomAllreduce` of the reference (distributed/device_communicators/custom_all_reduce.py:146-568:
constructor on a non-NCCL group, `should_custom_ar`, `custom_all_reduce`, `all_reduce`, `capture`, `close`)
with a simpler buffer protocol: one uncached region per rank, exported once through the IPC seam, and no
per-tensor or per-graph registration (the kernel stages its input into the region itself).

Ranks may live on different GPUs of one node (the regions are then read over xGMI) or share a GPU
(world_size-2 tests on the one-GPU box: same kernels, same flags, the "link" is HBM).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import logging
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from semi_pd_amd import _lib

logger = logging.getLogger(__name__)


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]
    # custom_all_reduce.py:148-151: "crossover is at 16MB buffer size for ROCm"
    _MAX_CAR_SIZE = 2 * 8192 * 1024

    def __init__(self, group: dist.ProcessGroup, device: torch.device, max_size: int = _MAX_CAR_SIZE,
                 capacity: Optional[int] = None) -> None:
        """max_size: largest all-reduce payload that takes the peer-memory kernel (above it RCCL is faster);
        capacity (>= max_size, default = max_size): payload bytes one slot of the shared region holds -- the expert-parallel
        all-to-all stages whole token chunks and wants more than an all-reduce ever should."""
        self.disabled = True
        self._comm = None
        self._region = None
        self._peer_bases: List[int] = []
        self._IS_CAPTURING = False
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, \
            "CustomAllreduce exchanges its handles on a CPU group (custom_all_reduce.py:181-183)"
        self.rank = dist.get_rank(group=group)
        self.world_size = dist.get_world_size(group=group)
        if self.world_size == 1 or self.world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        self.device = torch.device(device)
        self.max_size = int(max_size)
        self.capacity = max(int(capacity or 0), self.max_size)
        lib = _lib.load()

        def agree(ok: bool) -> bool:
            votes: List[Optional[bool]] = [None] * self.world_size
            dist.all_gather_object(votes, bool(ok), group=group)
            return all(votes)

        # Every phase ends in a vote on the CPU group: a rank that cannot allocate, export or map (no P2P route,
        # an IPC quirk of the platform) makes ALL ranks fall back to the backend collectives, instead of raising
        # on one rank while the others wait in a barrier.
        def fail_here(phase: str) -> None:
            # test hook (tests/test_gpu_all_reduce.py): SEMIPD_AR_TEST_FAIL="<phase>:<rank>" breaks one rank in one phase
            if os.environ.get("SEMIPD_AR_TEST_FAIL", "") == f"{phase}:{self.rank}":
                raise RuntimeError(f"injected failure in phase {phase!r}")

        mine = None
        try:
            fail_here("export")
            with torch.cuda.device(self.device):
                region_bytes = int(lib.semipd_ar_region_size(self.capacity))
                region = C.c_void_p()
                _lib.check(lib.semipd_ar_alloc_shared(region_bytes, C.addressof(region)), "ar_alloc_shared")
                self._region = region.value
                handle = (C.c_uint8 * 64)()
                offset = C.c_uint64()
                _lib.check(lib.semipd_ipc_get_handle(self._region, C.addressof(handle), C.addressof(offset)),
                           "ipc_get_handle")
                mine = (bytes(handle), int(offset.value), os.getpid())
        except Exception as e:  # noqa: BLE001
            logger.warning("peer-memory all-reduce: rank %d cannot export its region: %s", self.rank, e)
        everyone: List[Optional[tuple]] = [None] * self.world_size
        dist.all_gather_object(everyone, mine, group=group)
        if any(x is None for x in everyone):
            self.close()
            return
        ok = True
        try:
            fail_here("map")
            with torch.cuda.device(self.device):
                bases = []
                for r, (h, off, pid) in enumerate(everyone):
                    if r == self.rank:
                        bases.append(self._region)
                        continue
                    assert pid != os.getpid(), "one rank per process"
                    base = C.c_void_p()
                    hb = (C.c_uint8 * 64).from_buffer_copy(h)
                    _lib.check(lib.semipd_ipc_open(C.addressof(hb), self.device.index or 0, C.addressof(base)), "ipc_open")
                    self._peer_bases.append(base.value)
                    bases.append(base.value + off)
                arr = (C.c_void_p * self.world_size)(*bases)
                comm = C.c_void_p()
                _lib.check(lib.semipd_ar_init(C.addressof(arr), region_bytes, self.rank, self.world_size,
                                              C.addressof(comm)), "ar_init")
                self._comm = comm.value
        except Exception as e:  # noqa: BLE001
            logger.warning("peer-memory all-reduce: rank %d cannot map its peers: %s", self.rank, e)
            ok = False
        # (the vote is also the barrier: nobody starts reducing before every rank has mapped every region)
        if not agree(ok):
            self.close()
            return
        self.disabled = False
        if os.environ.get("SEMIPD_AR_SELF_TEST", "1") != "0" and not self._self_test():
            logger.warning("peer-memory all-reduce failed its start-up self-test on rank %d of %d; "
                           "falling back to the backend collectives", self.rank, self.world_size)
            self.close()

    def _self_test(self) -> bool:
        """One-stage, two-stage and all-gather calls on data every rank can predict, with bounded flag
        waits: a peer mapping that does not work (no P2P route, stale caching) shows up here as a wrong
        sum or a timed-out wait instead of as a hung GPU in the first forward pass.  All ranks take the
        same decision."""
        lib = _lib.load()
        ok = True
        try:
            if os.environ.get("SEMIPD_AR_TEST_FAIL", "") == f"selftest:{self.rank}":
                raise RuntimeError("injected failure in phase 'selftest'")
            _lib.check(lib.semipd_ar_set_timeout_ms(self._comm, 2000), "ar_set_timeout_ms")
            with torch.cuda.device(self.device):
                for numel in (8, 1 << 20):  # 16 B; 2 MB of bf16 (two-stage for more than 2 ranks)
                    if numel * 2 > self.max_size:
                        continue
                    i = torch.arange(numel, device=self.device, dtype=torch.int64)
                    mine = ((i * 7 + self.rank * 13) % 31).to(torch.bfloat16)
                    want = sum(((i * 7 + r * 13) % 31) for r in range(self.world_size)).to(torch.bfloat16)
                    for _ in range(3):  # both slots of the double buffer
                        ok = ok and torch.equal(self.all_reduce(mine), want)
                    gathered = self.all_gather(mine)
                    for r in range(self.world_size):
                        ok = ok and torch.equal(gathered[r], ((i * 7 + r * 13) % 31).to(torch.bfloat16))
                torch.cuda.synchronize(self.device)
                count = C.c_uint32()
                _lib.check(lib.semipd_ar_timed_out(self._comm, C.addressof(count)), "ar_timed_out")
                ok = ok and count.value == 0
        except Exception as e:  # noqa: BLE001
            logger.warning("peer-memory all-reduce self-test raised: %s", e)
            ok = False
        finally:
            lib.semipd_ar_set_timeout_ms(self._comm, 0)
        votes: List[Optional[bool]] = [None] * self.world_size
        dist.all_gather_object(votes, bool(ok), group=self.group)
        return all(votes)

    # ------------------------------------------------------------------ policy
    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """custom_all_reduce.py:447-488: multiples of 16 bytes, (weakly) contiguous, below max_size."""
        if self.disabled or not inp.is_cuda:
            return False
        if inp.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return False
        size = inp.numel() * inp.element_size()
        if size == 0 or size % 16 != 0 or not inp.is_contiguous() or inp.data_ptr() % 16 != 0:
            return False
        return size <= self.max_size

    def too_big_only(self, inp: torch.Tensor) -> bool:
        """A tensor should_custom_ar turns down for its size alone (all_reduce_in_pieces takes it)."""
        if self.disabled or not inp.is_cuda or inp.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            return False
        size = inp.numel() * inp.element_size()
        return size > self.max_size and size % 16 == 0 and inp.is_contiguous() and inp.data_ptr() % 16 == 0

    def all_reduce_in_pieces(self, inp: torch.Tensor) -> torch.Tensor:
        """In-place SUM of a tensor larger than max_size as consecutive peer-memory calls on pieces of at most max_size
        (element-wise reduce: cutting it changes no bit).  For an instance confined to a CU share: the kernels run on the
        CU-masked stream they are launched on, where the backend's own collectives would run on its internal, unmasked
        stream (distributed.py: set_comm_stream)."""
        flat = inp.view(-1)
        step = (self.max_size // 256) * 256 // inp.element_size()
        for a in range(0, flat.numel(), step):
            piece = flat[a:a + step]
            self.all_reduce(piece, out=piece)
        return inp

    def set_cu_trace(self, buf: Optional[torch.Tensor]) -> None:
        """tests: every block of every collective kernel marks buf[xcc * 256 + hardware CU id] (int32 [2048], zeroed)."""
        if buf is not None:
            assert buf.is_cuda and buf.dtype == torch.int32 and buf.numel() >= 2048 and buf.is_contiguous()
        _lib.check(_lib.load().semipd_ar_set_cu_trace(self._comm, _lib.ptr(buf) if buf is not None else None), "ar_set_cu_trace")

    # ------------------------------------------------------------------ calls
    def all_reduce(self, inp: torch.Tensor, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Out-of-place SUM (custom_all_reduce.py:503-527); `out=inp` reduces in place."""
        if out is None:
            out = torch.empty_like(inp)
        lib = _lib.load()
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        _lib.check(lib.semipd_ar_all_reduce(self._comm, _lib.ptr(inp), _lib.ptr(out), inp.numel(),
                                            _lib.dtype_code(inp.dtype), stream), "ar_all_reduce")
        return out

    def should_custom_ag(self, inp: torch.Tensor) -> bool:
        if self.disabled or not inp.is_cuda or not inp.is_contiguous():
            return False
        size = inp.numel() * inp.element_size()
        return size > 0 and size % 16 == 0 and inp.data_ptr() % 16 == 0 and size <= self.max_size

    def all_gather(self, inp: torch.Tensor) -> torch.Tensor:
        """[world, *inp.shape]: rank r's tensor at index r (any dtype: bytes are copied)."""
        out = torch.empty((self.world_size,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)
        lib = _lib.load()
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        _lib.check(lib.semipd_ar_all_gather(self._comm, _lib.ptr(inp), _lib.ptr(out), inp.numel() * inp.element_size(),
                                            stream), "ar_all_gather")
        return out

    # ------------------------------------------------------------------ expert-parallel all-to-all
    def ep_dispatch(self, x: torch.Tensor, topk_ids: torch.Tensor, topk_weights: torch.Tensor, experts_per_rank: int,
                    max_recv: int, check_overflow: Optional[bool] = None):
        """csrc/all_reduce.hip: semipd_ep_dispatch (oracle/ops.py: ep_dispatch).  x [T, H] (this rank's tokens), topk_ids
        [T, k] int32 GLOBAL expert ids, topk_weights [T, k] fp32.  Returns a state dict with the received rows (recv_x
        [max_recv, H], recv_expert, recv_weight, recv_count on the device) and what ep_combine needs to find the way back.
        max_recv must be the SAME on every rank.  Rows beyond it are dropped (and contribute zero in ep_combine): with
        check_overflow (default: outside graph capture, whenever max_recv is below world x T x k, the most this rank could
        receive from ranks with as many tokens) the call reads recv_count back and raises instead."""
        T, H = x.shape
        k = topk_ids.shape[1]
        assert x.is_contiguous() and topk_ids.dtype == torch.int32 and topk_weights.dtype == torch.float32
        assert topk_ids.is_contiguous() and topk_weights.is_contiguous()
        dev = x.device
        st = {"recv_x": torch.empty((max_recv, H), dtype=x.dtype, device=dev),
              "recv_expert": torch.zeros(max_recv, dtype=torch.int32, device=dev),
              "recv_weight": torch.zeros(max_recv, dtype=torch.float32, device=dev),
              "recv_count": torch.zeros(1, dtype=torch.int32, device=dev),
              "send_within": torch.empty((T, k), dtype=torch.int32, device=dev),
              "counts_all": torch.zeros((self.world_size, self.world_size), dtype=torch.int32, device=dev),
              "topk_ids": topk_ids, "experts_per_rank": int(experts_per_rank), "max_recv": int(max_recv), "tokens": T, "top_k": k}
        lib = _lib.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.semipd_ep_dispatch(self._comm, _lib.ptr(x), _lib.ptr(topk_ids), _lib.ptr(topk_weights), T, k,
                                          H * x.element_size(), int(experts_per_rank), _lib.ptr(st["recv_x"]),
                                          _lib.ptr(st["recv_expert"]), _lib.ptr(st["recv_weight"]), int(max_recv),
                                          _lib.ptr(st["recv_count"]), _lib.ptr(st["send_within"]), _lib.ptr(st["counts_all"]),
                                          stream), "ep_dispatch")
        if check_overflow is None:
            check_overflow = max_recv < self.world_size * T * k and not torch.cuda.is_current_stream_capturing()
        if check_overflow:
            got = int(st["recv_count"].item())
            if got > max_recv:
                raise RuntimeError(f"ep_dispatch: {got} rows routed to rank {self.rank}'s experts, max_recv = {max_recv}: "
                                   "rows were dropped (size max_recv for the worst case or check recv_count yourself)")
        return st

    def ep_combine(self, y: torch.Tensor, st: dict) -> torch.Tensor:
        """semipd_ep_combine: y [max_recv, H] = this rank's (weighted) expert outputs for the rows it received, in received
        order; returns [tokens, H]: the sum over each token's k entries in j order, fp32, one rounding."""
        H = y.shape[1]
        assert y.is_contiguous() and y.shape[0] >= st["max_recv"]
        out = torch.empty((st["tokens"], H), dtype=y.dtype, device=y.device)
        lib = _lib.load()
        stream = torch.cuda.current_stream(y.device).cuda_stream
        _lib.check(lib.semipd_ep_combine(self._comm, _lib.ptr(y), _lib.ptr(st["recv_count"]), st["max_recv"],
                                         _lib.ptr(st["topk_ids"]), _lib.ptr(st["send_within"]), _lib.ptr(st["counts_all"]),
                                         _lib.ptr(out), st["tokens"], st["top_k"], H, st["experts_per_rank"],
                                         _lib.dtype_code(y.dtype), stream), "ep_combine")
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """custom_all_reduce.py:529-552: None when the tensor does not qualify (the caller falls back to
        RCCL).  Nothing differs between eager and captured calls here."""
        if not self.should_custom_ar(input):
            return None
        return self.all_reduce(input)

    @contextlib.contextmanager
    def capture(self):
        """custom_all_reduce.py:369-381 registers the graph's buffers afterwards; nothing to register here."""
        self._IS_CAPTURING = True
        try:
            yield
        finally:
            self._IS_CAPTURING = False

    def close(self) -> None:
        if self._comm is None and self._region is None:
            return
        lib = _lib.load()
        if self._comm is not None:
            lib.semipd_ar_dispose(self._comm)
            self._comm = None
        for b in self._peer_bases:
            lib.semipd_ipc_close(b)
        self._peer_bases = []
        if self._region is not None:
            lib.semipd_ar_free_shared(self._region)
            self._region = None
        self.disabled = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
