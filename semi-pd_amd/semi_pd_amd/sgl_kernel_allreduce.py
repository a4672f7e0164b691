"""The reference's ROCm custom all-reduce op set, by name, on top of csrc/all_reduce.hip.

`sgl_kernel.allreduce` (sgl-kernel/python/sgl_kernel/allreduce.py:5-52, schemas in
sgl-kernel/csrc/torch_extension_rocm.cc:25-55) is what the reference's `CustomAllreduce`
(distributed/device_communicators/custom_all_reduce.py:283-302, 383-426, 490-501, 554-562) calls on ROCm:

    meta   = allocate_meta_buffer(meta_size() + max_size)            # uncached, shareable
    handle = get_meta_buffer_ipc_handle(meta)                        # exchanged on a CPU group
    fa     = init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink)
    register_buffer(fa, buffer, handles, offsets)                    # the staging buffer for unregistered inputs
    all_reduce_reg(fa, inp, out) / all_reduce_unreg(fa, inp, buffer, out)
    get_graph_buffer_ipc_meta(fa) ; register_graph_buffers(fa, handles, offsets)   # after a graph capture
    dispose(fa)

A maintainer who keeps the reference's `CustomAllreduce` class can point its `ops` at this module.  The kernel
underneath stages every input into its own rank's region (double buffered) before the peers read it, so it never reads
a caller's tensor across ranks: `register_buffer` and the two graph-buffer calls have nothing to register and are
accepted no-ops (`get_graph_buffer_ipc_meta` returns an empty list), `all_reduce_reg` and `all_reduce_unreg` are the
same staged reduction, and the meta buffer holds the signal block AND the staging area -- which is why
`allocate_meta_buffer(meta_size() + max_size)` returns `semipd_ar_region_size(max_size)` bytes, not the sum.
`semi_pd_amd/custom_all_reduce.py` is the build's own, shorter host path over the same C-ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from semi_pd_amd import _lib


class _Region:
    """Device memory from semipd_ar_alloc_shared exposed to torch without a copy."""

    def __init__(self, ptr: int, nbytes: int):
        self.ptr, self.nbytes = ptr, nbytes
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class _Comm:
    def __init__(self, handle: int, peers: List[int]):
        self.handle, self.peers = handle, peers


_regions: Dict[int, Tuple[_Region, int]] = {}   # data_ptr -> (region, max_bytes)
_comms: Dict[int, _Comm] = {}


def meta_size() -> int:
    """Bytes of the signal block at the head of a rank's region (custom_all_reduce.hip:117)."""
    return int(_lib.load().semipd_ar_meta_size())


def allocate_meta_buffer(size: int) -> torch.Tensor:
    """custom_all_reduce.hip:156-172: uncached device memory every peer can map.  `size` = meta_size() + max_size as in the
    reference's constructor; the region also carries this build's staging area (see the module docstring)."""
    lib = _lib.load()
    max_bytes = int(size) - meta_size()
    if max_bytes <= 0:
        raise RuntimeError(f"allocate_meta_buffer: size {size} leaves no room behind the {meta_size()}-byte signal block")
    region_bytes = int(lib.semipd_ar_region_size(max_bytes))
    ptr = C.c_void_p()
    _lib.check(lib.semipd_ar_alloc_shared(region_bytes, C.addressof(ptr)), "ar_alloc_shared")
    region = _Region(ptr.value, region_bytes)
    t = torch.as_tensor(region, device=torch.device("cuda", torch.cuda.current_device()))
    _regions[t.data_ptr()] = (region, max_bytes)
    return t


def free_meta_buffer(meta: torch.Tensor) -> None:
    """custom_all_reduce.hip:144 (a deleter the reference hands to torch): returns the region to the driver."""
    region, _ = _regions.pop(meta.data_ptr())
    _lib.load().semipd_ar_free_shared(C.c_void_p(region.ptr))


def get_meta_buffer_ipc_handle(inp: torch.Tensor) -> torch.Tensor:
    """custom_all_reduce.hip:146-154: the 64-byte hipIpcMemHandle of the allocation `inp` lives in, as a CPU uint8 tensor."""
    handle = (C.c_uint8 * 64)()
    offset = C.c_uint64()
    _lib.check(_lib.load().semipd_ipc_get_handle(C.c_void_p(inp.data_ptr()), C.addressof(handle), C.addressof(offset)),
               "ipc_get_handle")
    return torch.frombuffer(bytearray(bytes(handle)), dtype=torch.uint8)


def init_custom_ar(meta: torch.Tensor, rank_data: torch.Tensor, handles: List[bytes], offsets: List[int], rank: int,
                   full_nvlink: bool) -> int:
    """custom_all_reduce.hip:13-51.  `handles[r]` / `offsets[r]` locate rank r's meta buffer; the own entry is ignored
    (the own region is `meta`).  `rank_data` and `full_nvlink` are accepted for the signature: nothing is registered per
    tensor here, and the one- / two-stage choice is made by size and world size inside the kernel launch."""
    lib = _lib.load()
    world = len(handles)
    if len(offsets) != world or not (0 <= rank < world):
        raise RuntimeError("init_custom_ar: handles, offsets and rank disagree")
    if meta.data_ptr() not in _regions:
        raise RuntimeError("init_custom_ar: meta must come from allocate_meta_buffer")
    region, _ = _regions[meta.data_ptr()]
    device = meta.device.index if meta.device.index is not None else torch.cuda.current_device()
    bases, peers = [], []
    try:
        for r in range(world):
            if r == rank:
                bases.append(region.ptr)
                continue
            hb = (C.c_uint8 * 64).from_buffer_copy(bytes(handles[r]))
            base = C.c_void_p()
            _lib.check(lib.semipd_ipc_open(C.addressof(hb), device, C.addressof(base)), "ipc_open")
            peers.append(base.value)
            bases.append(base.value + int(offsets[r]))
        arr = (C.c_void_p * world)(*bases)
        comm = C.c_void_p()
        _lib.check(lib.semipd_ar_init(C.addressof(arr), region.nbytes, rank, world, C.addressof(comm)), "ar_init")
    except Exception:
        for b in peers:
            lib.semipd_ipc_close(C.c_void_p(b))
        raise
    _comms[comm.value] = _Comm(comm.value, peers)
    return comm.value


def _reduce(fa: int, inp: torch.Tensor, out: torch.Tensor, what: str) -> None:
    if fa not in _comms:
        raise RuntimeError(f"{what}: unknown communicator {fa}")
    if inp.dtype not in (torch.float32, torch.float16, torch.bfloat16) or out.dtype != inp.dtype or out.numel() != inp.numel():
        raise RuntimeError(f"{what}: float32 / float16 / bfloat16 tensors of one size expected")
    if not (inp.is_contiguous() and out.is_contiguous()):
        raise RuntimeError(f"{what}: contiguous tensors expected (custom_all_reduce.py:139-145)")
    _lib.check(_lib.load().semipd_ar_all_reduce(C.c_void_p(fa), _lib.ptr(inp), _lib.ptr(out), inp.numel(), _lib.dtype_code(inp.dtype),
                                                _lib.current_stream(inp.device)), what)


def all_reduce_reg(fa: int, inp: torch.Tensor, out: torch.Tensor) -> None:
    """custom_all_reduce.hip:89-95."""
    _reduce(fa, inp, out, "all_reduce_reg")


def all_reduce_unreg(fa: int, inp: torch.Tensor, reg_buffer: torch.Tensor, out: torch.Tensor) -> None:
    """custom_all_reduce.hip:97-110 copies `inp` into the registered `reg_buffer` first; the staging copy is part of
    this build's kernel, `reg_buffer` is not touched."""
    if reg_buffer.numel() * reg_buffer.element_size() < inp.numel() * inp.element_size():
        raise RuntimeError("all_reduce_unreg: registered buffer is too small for the input (custom_all_reduce.hip:105-106)")
    _reduce(fa, inp, out, "all_reduce_unreg")


def register_buffer(fa: int, t: torch.Tensor, handles: List[bytes], offsets: List[int]) -> None:
    """custom_all_reduce.hip:119-124.  Nothing to register: peers never read a caller's tensor (module docstring)."""
    if fa not in _comms:
        raise RuntimeError(f"register_buffer: unknown communicator {fa}")
    if len(handles) != len(offsets):
        raise RuntimeError("register_buffer: handles and offsets disagree")


def get_graph_buffer_ipc_meta(fa: int) -> Tuple[torch.Tensor, List[int]]:
    """custom_all_reduce.hip:126-136: the addresses a graph capture recorded.  None here: (empty handle tensor, [])."""
    if fa not in _comms:
        raise RuntimeError(f"get_graph_buffer_ipc_meta: unknown communicator {fa}")
    return torch.empty(0, dtype=torch.uint8), []


def register_graph_buffers(fa: int, handles: List[bytes], offsets: List[List[int]]) -> None:
    """custom_all_reduce.hip:138-142.  Accepted for the call sequence; see get_graph_buffer_ipc_meta."""
    if fa not in _comms:
        raise RuntimeError(f"register_graph_buffers: unknown communicator {fa}")


def dispose(fa: int) -> None:
    """custom_all_reduce.hip:112-115; also unmaps the peers' regions."""
    comm = _comms.pop(fa, None)
    if comm is None:
        return
    lib = _lib.load()
    lib.semipd_ar_dispose(C.c_void_p(fa))
    for b in comm.peers:
        lib.semipd_ipc_close(C.c_void_p(b))
