"""Mirror of python/sglang/semi_pd/utils.py (IPCInfo, InstanceRole, AggregatedSocket, dtype table,
compute-share knobs) for the MI355X build.

The reference partitions the GPU with CUDA MPS (env CUDA_MPS_ACTIVE_THREAD_PERCENTAGE set before
each fork, entrypoints/engine.py:591-593, 632-634) and has no AMD path.  Here the same two knobs
    SEMI_PD_PREFILL_SM_PERCENTILE / SEMI_PD_DECODE_SM_PERCENTILE   (semi_pd/utils.py:10-11)
become CU masks: `cu_mask_env()` yields the environment for a child process (process-wide mask,
covers every stream, hipGraph and RCCL kernel of that process), `cu_masked_stream()` builds a
hipExtStreamCreateWithCUMask stream inside a process.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from enum import Enum
from typing import Dict, List

import torch

import semi_pd_ipc
from semi_pd_amd import _lib

# The reference's defaults (semi_pd/utils.py:10-11) are prefill 80 % of the SMs, decode 100 %: overlapping shares.  With CU masks
# that is a NESTED pair: the prefill instance keeps to the lowest CUs, the decode instance may use every CU and has the rest to
# itself.  Shares are whole groups of 32 CUs (one per shader engine of every XCD, csrc/ipc.hip: semipd_cu_mask_fill), so on
# an MI355X the choice is between 192 CUs (75 .. 81 %) and 224 (82 .. 93 %); measured, Llama-3-8B at 32 req/s (DESIGN.md 4.4,
# 4.5): 192 CUs TTFT p50 40.7 ms / TBT p99 11.6 ms, 224 CUs 29.4 / 15.6 -- and 224 CUs with the decode-step deadline of
# semi_pd/step_pacer.py 33.6 / 11.9.  The default here is therefore 88 % (224 CUs) where the reference has 80; the
# environment variables of the reference still override it.
PREFILL_ENGINE_SM_PERCENTILE = int(os.getenv("SEMI_PD_PREFILL_SM_PERCENTILE", 88))
DECODE_ENGINE_SM_PERCENTILE = int(os.getenv("SEMI_PD_DECODE_SM_PERCENTILE", 100))


@dataclass
class IPCInfo:
    """Same fields as the reference dataclass (semi_pd/utils.py:14-23)."""
    params_info: dict
    weight_handles: dict
    register_buffer_handles: dict
    kv_cache_handles: list
    kvcache_info: dict
    req_to_token_handle: list
    req_to_token_info: dict


class InstanceRole(Enum):
    PREFILL = 0
    DECODE = 1
    OTHER = 2


class AggregatedSocket:
    """Fan-out sender (semi_pd/utils.py:31-37): the tokenizer sends every request to D first, then P."""

    def __init__(self, sockets: List):
        self.sockets = sockets

    def send_pyobj(self, obj):
        for socket in self.sockets:
            socket.send_pyobj(obj)


DTYPE_TO_ATEN = {v: k for k, v in semi_pd_ipc.ATEN_TO_DTYPE.items()}


def get_ipc_handle(tensor: torch.Tensor):
    """(handle, storage_offset_bytes) like the reference helper (semi_pd/utils.py:66-76) — the offset
    comes from hipMemGetAddressRange instead of storage()._share_cuda_()."""
    return semi_pd_ipc.get_ipc_handle_and_offset(tensor)


def convert_ipc_handle_to_tensor(ipc_handle, size, dtype, device):
    return semi_pd_ipc.convert_ipc_handle_to_tensor(ipc_handle, size, DTYPE_TO_ATEN[dtype], device)


def get_device_sm_count(rank: int = 0):
    return semi_pd_ipc.get_device_sm_count(rank)


# --------------------------------------------------------------------------- CU masks
def cu_mask_words(num_cus: int, percent: int, from_top: bool) -> List[int]:
    lib = _lib.load()
    words = (num_cus + 31) // 32
    buf = (C.c_uint32 * words)()
    n = lib.semipd_cu_mask_fill(num_cus, int(percent), 1 if from_top else 0, C.addressof(buf), words)
    if n <= 0:
        raise RuntimeError(f"cu_mask_fill failed: {_lib.last_error()}")
    return [int(w) for w in buf]


def cu_mask_env(gpu_id: int, num_cus: int, percent: int, from_top: bool, library_grid: bool = False) -> Dict[str, str]:
    """Environment that confines a *process* to `percent` of the CUs of `gpu_id`.
    ROCr reads HSA_CU_MASK ("<gpu>:<cu list>") when the process creates its queues; a 100 % share
    needs no mask.  `library_grid` also tells hipBLASLt's stream-K GEMMs (one persistent workgroup per CU
    of the DEVICE by default, i.e. two rounds under any mask) how many CUs the process really owns."""
    if percent >= 100:
        return {}
    words = cu_mask_words(num_cus, percent, from_top)
    bits = [i for i in range(num_cus) if words[i >> 5] >> (i & 31) & 1]
    extra = {"TENSILE_STREAMK_MAX_CUS": str(len(bits))} if library_grid else {}
    # compress into ranges
    ranges, start, prev = [], bits[0], bits[0]
    for b in bits[1:]:
        if b != prev + 1:
            ranges.append((start, prev))
            start = b
        prev = b
    ranges.append((start, prev))
    spec = ",".join(f"{a}-{b}" if a != b else f"{a}" for a, b in ranges)
    return {"HSA_CU_MASK": f"{gpu_id}:{spec}", **extra}


def cu_masked_stream(device_index: int, percent: int, from_top: bool) -> torch.cuda.Stream:
    """torch stream backed by hipExtStreamCreateWithCUMask (semipd_stream_create_cu_mask)."""
    lib = _lib.load()
    n = get_device_sm_count(device_index)
    words = cu_mask_words(n, percent, from_top)
    arr = (C.c_uint32 * len(words))(*words)
    s = C.c_void_p(0)
    _lib.check(lib.semipd_stream_create_cu_mask(device_index, C.addressof(arr), len(words), C.addressof(s)),
               "stream_create_cu_mask")
    return torch.cuda.ExternalStream(int(s.value), device=torch.device("cuda", device_index))
