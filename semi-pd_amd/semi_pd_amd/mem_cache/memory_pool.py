"""Request/token/KV pools of the Semi-PD hot path.

Same roles and method names as the reference pools (mem_cache/memory_pool.py:46-96 ReqToTokenPool,
:124-184 TokenToKVPoolAllocator, :187-346 MHATokenToKVPool, :379-452 MLATokenToKVPool), with the
layout chosen for one 288 GB HBM3E device:

 * all layers' K and V rows live in ONE slab  kv[L, 2, N+1, Hkv, D]  (MLA: kv[L, N+1, 1, 576]);
   per-layer buffers are views, so a single hipIpcMemHandle covers the whole cache and the
   prefill process maps it once (bypass_create_buffers=True leaves the views empty until
   share_params_from_ipc fills them, model_runner.py:563-624);
 * page size is 1 (token-granular), slot 0 is the dummy slot for padded tokens
   (memory_pool.py:172-176);
 * the free list stays on the host: the decode process is the only allocator
   (semi_pd_decode_scheduler.py:166-337) and never needs it on the device.
"""
from __future__ import annotations

import collections
from typing import List, Optional, Tuple, Union

import torch

from semi_pd_amd import ops


class ReqToTokenPool:
    """Request slot -> KV-pool slots of its tokens: the int32 table [size, max_context_len] both instances map
    (interface of mem_cache/memory_pool.py:46-96: `req_to_token`, write / available_size / alloc / free / clear).

    The slot allocator behind it is this build's own: a FIFO deque plus an in-use bitmap.  Only the decode instance
    allocates (the prefill instance builds the pool with `bypass_create_buffers` and maps the table through IPC), a
    slot is handed out at most once until it comes back, and releasing a slot twice or releasing one that was never
    handed out raises instead of silently putting a duplicate on the free list (two requests would then share a row
    of the table)."""

    def __init__(self, size: int, max_context_len: int, device: str, bypass_create_buffers: bool = False):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token: Optional[torch.Tensor] = None
        if not bypass_create_buffers:
            self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.clear()

    def clear(self):
        self._free = collections.deque(range(self.size))
        self._in_use = bytearray(self.size)

    def available_size(self) -> int:
        return len(self._free)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self._free):
            return None
        slots = [self._free.popleft() for _ in range(need_size)]
        for i in slots:
            self._in_use[i] = 1
        return slots

    def free(self, free_index: Union[int, List[int]]):
        for i in ((free_index,) if isinstance(free_index, int) else free_index):
            i = int(i)
            if not (0 <= i < self.size) or not self._in_use[i]:
                raise ValueError(f"ReqToTokenPool.free: request slot {i} is not allocated")
            self._in_use[i] = 0
            self._free.append(i)

    def write(self, indices, values):
        self.req_to_token[indices] = values


class TokenToKVPoolAllocator:
    """Free list of KV slots 1..size (int64, FIFO like the reference's tensor slicing,
    mem_cache/memory_pool.py:130-190).  The list is kept as a deque of chunks: allocation takes from the
    head, a free appends one chunk at the tail, so neither is O(pool size) (the reference re-concatenates
    the whole free tensor on every free)."""

    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache):
        self.size = size
        self.dtype = dtype
        self.device = device
        self.page_size = 1
        self._kvcache = kvcache
        self.is_not_in_free_group = True
        self.free_group: List[torch.Tensor] = []
        self.clear()

    def available_size(self):
        return self._avail

    def get_kvcache(self):
        return self._kvcache

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > self._avail:
            return None
        parts, need = [], int(need_size)
        while need > 0:
            head = self._chunks[0]
            if head.numel() <= need:
                parts.append(self._chunks.popleft())
                need -= head.numel()
            else:
                parts.append(head[:need])
                self._chunks[0] = head[need:]
                need = 0
        self._avail -= int(need_size)
        if not parts:
            return torch.empty(0, dtype=torch.int64)
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def free(self, free_index):
        """free_index: a list of ints or a tensor (a device tensor is copied to the host, which waits for the
        stream: the schedulers pass host data)."""
        if not torch.is_tensor(free_index):
            free_index = torch.tensor(list(free_index), dtype=torch.int64)
        if free_index.numel() == 0:
            return
        free_index = free_index.to("cpu", torch.int64)
        if self.is_not_in_free_group:
            self._chunks.append(free_index)
            self._avail += free_index.numel()
        else:
            self.free_group.append(free_index)

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.concat(self.free_group))

    @property
    def free_slots(self) -> torch.Tensor:
        """The free list in allocation order (diagnostics / tests)."""
        return torch.cat(list(self._chunks)) if self._chunks else torch.empty(0, dtype=torch.int64)

    def clear(self):
        # slot 0 is reserved for dummy writes of padded tokens
        self._chunks = collections.deque([torch.arange(1, self.size + 1, dtype=torch.int64)])
        self._avail = self.size
        self.is_not_in_free_group = True
        self.free_group = []


def ipc_safe_zeros(shape, dtype: torch.dtype, device) -> torch.Tensor:
    """torch.zeros(shape, dtype) inside an allocation another process can import: on this ROCm
    hipIpcOpenMemHandle hangs when the allocation size modulo 4 GiB is >= 2 GiB (see csrc/ipc.hip), so
    the backing allocation is padded up to the next multiple of 4 GiB in that case (< 2 GiB of 288).
    The caching allocator rounds large requests up to 2 MiB, which is what the check looks at."""
    numel = 1
    for d in shape:
        numel *= int(d)
    nbytes = numel * dtype.itemsize
    alloc = -(-nbytes // (2 << 20)) * (2 << 20)
    if (alloc & 0xFFFFFFFF) >= (1 << 31):
        alloc = -(-alloc // (1 << 32)) * (1 << 32)
    raw = torch.zeros(alloc, dtype=torch.uint8, device=device)
    return raw[:nbytes].view(dtype).view(*shape)


class MHATokenToKVPool:
    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int,
                 layer_num: int, device: str, bypass_create_buffers: bool = False):
        assert page_size == 1, "Semi-PD runs with token-granular pages (schedule_batch.py:937)"
        self.size, self.page_size, self.dtype = size, page_size, dtype
        self.head_num, self.head_dim, self.layer_num, self.device = head_num, head_dim, layer_num, device
        self.k_buffer: List[torch.Tensor] = []
        self.v_buffer: List[torch.Tensor] = []
        self.slab: Optional[torch.Tensor] = None
        if not bypass_create_buffers:
            self._create_buffers()

    def _create_buffers(self):
        # [L, 2, N+page, Hkv, D]: one allocation, one IPC handle
        self.slab = ipc_safe_zeros((self.layer_num, 2, self.size + self.page_size, self.head_num, self.head_dim),
                                   self.dtype, self.device)
        self.k_buffer = [self.slab[i, 0] for i in range(self.layer_num)]
        self.v_buffer = [self.slab[i, 1] for i in range(self.layer_num)]

    def get_kv_size_bytes(self):
        n = (self.size + self.page_size) * self.head_num * self.head_dim * self.dtype.itemsize * self.layer_num
        return n, n

    def get_key_buffer(self, layer_id: int):
        return self.k_buffer[layer_id]

    def get_value_buffer(self, layer_id: int):
        return self.v_buffer[layer_id]

    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.k_buffer[layer_id], self.v_buffer[layer_id]

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor):
        layer_id = layer.layer_id
        # an fp8 pool (--kv-cache-dtype fp8_e5m2 / fp8_e4m3, memory_pool.py:205-209, 326-336) converts
        # inside the scatter kernel; any other mismatch is converted here like the reference does
        if cache_k.dtype != self.dtype and self.dtype not in (torch.float8_e5m2, torch.float8_e4m3fn):
            cache_k, cache_v = cache_k.to(self.dtype), cache_v.to(self.dtype)
        ops.store_kv_rows(self.k_buffer[layer_id], loc, cache_k.view(-1, self.head_num, self.head_dim))
        ops.store_kv_rows(self.v_buffer[layer_id], loc, cache_v.view(-1, self.head_num, self.head_dim))


class MLATokenToKVPool:
    """Latent KV rows [N+page, 1, kv_lora_rank + qk_rope_head_dim] per layer; the value buffer is
    the first kv_lora_rank columns of the same rows (memory_pool.py:379-452)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, kv_lora_rank: int,
                 qk_rope_head_dim: int, layer_num: int, device: str, bypass_create_buffers: bool = False):
        assert page_size == 1
        self.size, self.page_size, self.dtype = size, page_size, dtype
        self.kv_lora_rank, self.qk_rope_head_dim = kv_lora_rank, qk_rope_head_dim
        self.layer_num, self.device = layer_num, device
        self.kv_buffer: List[torch.Tensor] = []
        self.slab: Optional[torch.Tensor] = None
        if not bypass_create_buffers:
            self.slab = ipc_safe_zeros((layer_num, size + page_size, 1, kv_lora_rank + qk_rope_head_dim), dtype,
                                       device)
            self.kv_buffer = [self.slab[i] for i in range(layer_num)]

    def get_key_buffer(self, layer_id: int):
        return self.kv_buffer[layer_id]

    def get_value_buffer(self, layer_id: int):
        return self.kv_buffer[layer_id][..., : self.kv_lora_rank]

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor = None):
        ops.store_kv_rows(self.kv_buffer[layer.layer_id], loc,
                          cache_k.view(-1, 1, self.kv_lora_rank + self.qk_rope_head_dim))
