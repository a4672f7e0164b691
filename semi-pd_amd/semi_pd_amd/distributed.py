"""Tensor-parallel group state and collectives for the hot path.

One process per GPU; device collectives go through torch.distributed's "nccl" backend (= RCCL over
xGMI on ROCm), scheduler broadcasts through a gloo CPU group — the same split as the reference
(distributed/parallel_state.py:376-489 GroupCoordinator.all_reduce / all_gather,
distributed/communication_op.py:11-21, managers/scheduler.py:645-659 broadcast_pyobj).
Prefill ranks and decode ranks form two independent worlds (separate ports, scheduler.py:249-259).

On CPU tensors the same code runs over gloo, which is how tests cover world_size 2 without GPUs.
"""
from __future__ import annotations

import os
import pickle
from typing import Any, List, Optional

import torch
import torch.distributed as dist

_TP_RANK = 0
_TP_SIZE = 1
_DEVICE_GROUP: Optional[dist.ProcessGroup] = None
_CPU_GROUP: Optional[dist.ProcessGroup] = None
_CUSTOM_AR = None  # custom_all_reduce.CustomAllreduce, GPU ranks of one node only


def init_distributed_environment(world_size: int, rank: int, distributed_init_method: str,
                                 backend: str = "nccl", device: Optional[torch.device] = None,
                                 timeout_s: int = 600, use_custom_all_reduce: bool = True,
                                 peer_region_capacity: Optional[int] = None) -> None:
    """model_runner.py:285-344 init_torch_distributed: one device group + one gloo group, and the
    peer-memory all-reduce on top of the gloo group (parallel_state.py:258-266 creates CustomAllreduce
    on the cpu_group unless --disable-custom-all-reduce)."""
    global _TP_RANK, _TP_SIZE, _DEVICE_GROUP, _CPU_GROUP, _CUSTOM_AR
    _TP_RANK, _TP_SIZE = rank, world_size
    if world_size == 1:
        _DEVICE_GROUP = _CPU_GROUP = None
        return
    import datetime
    if not dist.is_initialized():
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        # Under torchrun every descendant inherits TORCHELASTIC_USE_AGENT_STORE=True, and torch then makes NO rank the
        # server of a tcp:// rendezvous (it expects the launcher's agent store at that address): the scheduler
        # processes of a `torch.distributed.run ... bench.py --gpus N` job all sat in connect() until the timeout.
        # The groups here bring their own address and port, so rank 0 must serve it.
        agent_store = os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None) \
            if str(distributed_init_method).startswith("tcp://") else None
        try:
            dist.init_process_group(backend=backend, init_method=distributed_init_method, world_size=world_size,
                                    rank=rank, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
        finally:
            if agent_store is not None:
                os.environ["TORCHELASTIC_USE_AGENT_STORE"] = agent_store
    _DEVICE_GROUP = dist.group.WORLD
    _CPU_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s)) \
        if backend != "gloo" else dist.group.WORLD
    _CUSTOM_AR = None
    if use_custom_all_reduce and device is not None and torch.device(device).type == "cuda" \
            and os.environ.get("SEMIPD_DISABLE_CUSTOM_ALL_REDUCE", "0") != "1":
        from semi_pd_amd.custom_all_reduce import CustomAllreduce
        ar = CustomAllreduce(_CPU_GROUP, torch.device(device), capacity=peer_region_capacity)
        _CUSTOM_AR = None if ar.disabled else ar


def destroy_distributed_environment() -> None:
    global _TP_RANK, _TP_SIZE, _DEVICE_GROUP, _CPU_GROUP, _CUSTOM_AR
    if _CUSTOM_AR is not None:
        _CUSTOM_AR.close()
        _CUSTOM_AR = None
    if dist.is_initialized():
        dist.destroy_process_group()
    _TP_RANK, _TP_SIZE, _DEVICE_GROUP, _CPU_GROUP = 0, 1, None, None


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tp_cpu_group():
    return _CPU_GROUP


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of [T, hidden] after o_proj / down_proj / experts (layers/linear.py:1266)."""
    if _TP_SIZE == 1:
        return input_
    if _CUSTOM_AR is not None and _CUSTOM_AR.should_custom_ar(input_):
        # parallel_state.py:395-410: the peer-memory kernel first, RCCL for what it does not take
        return _CUSTOM_AR.all_reduce(input_, out=input_)
    if _CUSTOM_AR is not None and _CONFINED["on"]:
        # an instance on its CU share: EVERY payload stays on the peer-memory kernels, on the (masked) stream of the caller --
        # RCCL would launch on its own unmasked stream, i.e. on the CUs the policy keeps for the other instance (the
        # reference's MPS percentage confines NCCL too: engine.py:591-593)
        return _confined_all_reduce(input_)
    dist.all_reduce(input_, group=_DEVICE_GROUP)
    return input_


CONFINED_STATS = {"in_pieces": 0, "staged": 0}


def _confined_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """In-place SUM through the peer-memory kernels whatever the tensor looks like: above their size limit piece by piece
    (all_reduce_in_pieces); not contiguous, not a multiple of 16 bytes or misaligned: through a contiguous staging buffer
    padded with zeros to 16 bytes (the reduce is element-wise: the bits of the direct call); a dtype the kernels do not sum
    is refused -- nothing falls through to the backend's own, unconfined stream (round-5 verdict, multi-GPU item)."""
    ar = _CUSTOM_AR
    if ar.too_big_only(input_):
        CONFINED_STATS["in_pieces"] += 1
        return ar.all_reduce_in_pieces(input_)
    if input_.dtype not in (torch.float32, torch.bfloat16, torch.float16) or input_.numel() == 0:
        if input_.numel() == 0:
            return input_
        raise RuntimeError(f"all-reduce of a {input_.dtype} tensor on a CU-confined instance: the peer-memory kernels sum "
                           "fp32 / bf16 / f16 only and the backend's collective would leave the instance's CU share "
                           "(--cu-mask-mode env or none lifts the confinement)")
    CONFINED_STATS["staged"] += 1
    n, per16 = input_.numel(), 16 // input_.element_size()
    buf = torch.zeros(-(-n // per16) * per16, dtype=input_.dtype, device=input_.device)
    buf[:n].copy_(input_.reshape(-1))
    if ar.should_custom_ar(buf):
        ar.all_reduce(buf, out=buf)
    else:
        ar.all_reduce_in_pieces(buf)
    input_.copy_(buf[:n].view(input_.shape))
    return input_


def get_custom_all_reduce():
    return _CUSTOM_AR


# ---------------------------------------------------------------------------- all-reduce overlapped with the GEMMs
# The reference reduces on the compute stream, blocking (layers/linear.py:1266).  A prefill-sized row-parallel
# layer here is cut into token chunks: the all-reduce of chunk i runs on a communication stream while the GEMM of
# chunk i + 1 runs on the compute stream (RowParallelLinear.forward).  The reduce is element-wise, so the chunking
# does not change a single bit of the result.
_COMM_STREAMS = {}      # device index -> the communication stream in use (set_comm_stream; lazily a plain created stream)
_COMM_PLAIN = {}        # device index -> the plain (unmasked) created stream, kept for when the instance is on the whole chip
_CONFINED = {"on": False}


def set_comm_stream(device_index: int, stream, confined: bool = False) -> None:
    """model_executor/cu_share.py: the communication stream follows the compute stream.  While the instance runs on its CU
    share `stream` carries the SAME CU mask (the all-reduce overlapped with the GEMMs must not spill onto the CUs the
    policy keeps for the other instance) and `confined` routes every all-reduce through the peer-memory kernels; on the
    whole chip (`stream` None) a plain created stream is used."""
    if stream is None:
        _COMM_STREAMS.pop(device_index, None)
    else:
        _COMM_STREAMS[device_index] = stream
    _CONFINED["on"] = bool(confined)


def _comm_stream(dev: torch.device):
    st = _COMM_STREAMS.get(dev.index)
    if st is None:
        st = _COMM_PLAIN.get(dev.index)
        if st is None:
            st = _COMM_PLAIN[dev.index] = torch.cuda.Stream(device=dev)
    return st


OVERLAP_MIN_TOKENS = int(os.environ.get("SEMIPD_AR_OVERLAP_MIN_TOKENS", "1024"))   # below: one blocking call
_OVERLAP = {"enabled": os.environ.get("SEMIPD_DISABLE_AR_OVERLAP", "0") != "1"}
# what the overlapped path did in this process (reported with the scheduler's stats): chunk reduces issued next to a
# GEMM, and how many of them ran as peer-memory kernels on the communication stream
OVERLAP_STATS = {"overlapped_reduces": 0, "overlapped_reduces_peer_memory_kernel": 0}


def set_all_reduce_overlap(enabled: bool) -> None:
    _OVERLAP["enabled"] = bool(enabled)


def all_reduce_overlap_chunks(num_tokens: int) -> int:
    """Token chunks of a row-parallel layer whose all-reduce is overlapped with its own GEMM: chunks of at least
    512 rows (the GEMM stays efficient), at most 4 (each chunk costs one collective launch)."""
    if _TP_SIZE == 1 or not _OVERLAP["enabled"] or num_tokens < OVERLAP_MIN_TOKENS:
        return 1
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return 1
    return max(1, min(4, num_tokens // 512))


class _Pending:
    """A reduce in flight; wait() orders the CURRENT stream (or the host, on CPU tensors) behind it."""

    def __init__(self, work=None, event=None):
        self.work, self.event = work, event

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)


def tensor_model_parallel_all_reduce_async(input_: torch.Tensor) -> _Pending:
    """In-place SUM all-reduce of `input_` that does not block the compute stream: the peer-memory kernels run on
    a dedicated communication stream behind the work already queued on the current one, RCCL (and gloo on CPU)
    through async_op.  Every rank must issue the same sequence of calls; the caller waits on the returned handle
    before it reads `input_` or issues a blocking collective."""
    if _TP_SIZE == 1:
        return _Pending()
    OVERLAP_STATS["overlapped_reduces"] += 1
    if _CUSTOM_AR is not None and _CUSTOM_AR.should_custom_ar(input_):
        OVERLAP_STATS["overlapped_reduces_peer_memory_kernel"] += 1
        dev = input_.device
        comm = _comm_stream(dev)
        comm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(comm):
            _CUSTOM_AR.all_reduce(input_, out=input_)
            ev = comm.record_event()
        input_.record_stream(comm)
        return _Pending(event=ev)
    if _CUSTOM_AR is not None and _CONFINED["on"]:
        OVERLAP_STATS["overlapped_reduces_peer_memory_kernel"] += 1
        dev = input_.device
        comm = _comm_stream(dev)
        comm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(comm):
            _confined_all_reduce(input_)
            ev = comm.record_event()
        input_.record_stream(comm)
        return _Pending(event=ev)
    return _Pending(work=dist.all_reduce(input_, group=_DEVICE_GROUP, async_op=True))


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """All-gather along `dim` (logits [B, V/tp] -> [B, V], layers/logits_processor.py:426-427;
    parallel_state.py:438-489)."""
    if _TP_SIZE == 1:
        return input_
    if dim < 0:
        dim += input_.dim()
    input_ = input_.contiguous()
    if _CUSTOM_AR is not None and _CUSTOM_AR.should_custom_ag(input_):
        out = _CUSTOM_AR.all_gather(input_)
    elif _CUSTOM_AR is not None and _CONFINED["on"] and input_.numel() > 0:
        out = _confined_all_gather(input_)      # a confined instance: never the backend's own (unmasked) stream
    else:
        out = _gather_with_backend(input_)
    out = out.movedim(0, dim)
    shape = list(input_.shape)
    shape[dim] = shape[dim] * _TP_SIZE
    return out.reshape(shape)


def _confined_all_gather(input_: torch.Tensor) -> torch.Tensor:
    """[world, *input_.shape] through the peer-memory all-gather for a payload it turns down (above its size limit, not a
    multiple of 16 bytes): pieces of at most max_size of a flat copy padded to 16 bytes (bytes are copied: any dtype)."""
    ar = _CUSTOM_AR
    flat = input_.reshape(-1)
    n, es = flat.numel(), input_.element_size()
    per16 = max(1, 16 // es)
    padded = -(-n // per16) * per16
    if padded != n or flat.data_ptr() % 16:
        buf = torch.zeros(padded, dtype=input_.dtype, device=input_.device)
        buf[:n].copy_(flat)
    else:
        buf = flat
    out = torch.empty((_TP_SIZE, padded), dtype=input_.dtype, device=input_.device)
    step = max(per16, (ar.max_size // 256) * 256 // es)
    for a in range(0, padded, step):
        out[:, a:a + step] = ar.all_gather(buf[a:a + step])
    CONFINED_STATS["gathered_in_pieces"] = CONFINED_STATS.get("gathered_in_pieces", 0) + 1
    return out[:, :n].reshape((_TP_SIZE,) + tuple(input_.shape))


def _gather_with_backend(input_: torch.Tensor) -> torch.Tensor:
    out = torch.empty((_TP_SIZE,) + tuple(input_.shape), dtype=input_.dtype, device=input_.device)
    if dist.get_backend(_DEVICE_GROUP) == "nccl":
        dist.all_gather_into_tensor(out, input_, group=_DEVICE_GROUP)
    else:
        dist.all_gather(list(out.unbind(0)), input_, group=_DEVICE_GROUP)
    return out


def broadcast_pyobj(data: List[Any], rank: int, group, src: int = 0) -> List[Any]:
    """Pickle broadcast on the CPU group (utils.broadcast_pyobj, scheduler.py:645-659)."""
    if group is None or _TP_SIZE == 1:
        return data
    if rank == src:
        payload = pickle.dumps(data)
        size = torch.tensor([len(payload)], dtype=torch.long)
        dist.broadcast(size, src=src, group=group)
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        dist.broadcast(buf, src=src, group=group)
        return data
    size = torch.tensor([0], dtype=torch.long)
    dist.broadcast(size, src=src, group=group)
    buf = torch.empty(int(size.item()), dtype=torch.uint8)
    dist.broadcast(buf, src=src, group=group)
    return pickle.loads(bytes(buf.numpy()))


def all_ranks_agree(flag: bool) -> bool:
    """True when `flag` is true on EVERY tensor-parallel rank (one MIN all-reduce on the CPU group; start-up paths only).
    What a rank does per step must be a function of things all ranks share -- the step count, the batch, and whatever they
    agreed on here: host state that only one rank has (a failed graph capture, a full sample list) must never pick between
    two launch sequences whose peer-memory collectives differ (commit 19e5420: an N = 2 run hung on exactly that)."""
    if _CPU_GROUP is None or _TP_SIZE == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=_CPU_GROUP)
    return bool(t.item())


def barrier_cpu() -> None:
    if _CPU_GROUP is not None and _TP_SIZE > 1:
        dist.barrier(group=_CPU_GROUP)
