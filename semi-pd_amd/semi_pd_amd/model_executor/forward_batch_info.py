"""ForwardMode / ForwardBatch: the per-forward device-side description of a batch
(mirrors model_executor/forward_batch_info.py:226-332 `ForwardBatch.init_new`, positions :393-466)."""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum, auto
from typing import List, Optional

import torch

from semi_pd_amd import ops


class ForwardMode(IntEnum):
    EXTEND = auto()
    DECODE = auto()
    IDLE = auto()

    def is_extend(self):
        return self == ForwardMode.EXTEND

    def is_decode(self):
        return self == ForwardMode.DECODE

    def is_idle(self):
        return self == ForwardMode.IDLE


@dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    batch_size: int
    input_ids: torch.Tensor            # int64 [T]
    req_pool_indices: torch.Tensor     # int64 [B]
    seq_lens: torch.Tensor             # int64 [B]   (prefix + extend, or current length for decode)
    out_cache_loc: torch.Tensor        # int64 [T]
    seq_lens_sum: int
    positions: Optional[torch.Tensor] = None          # int64 [T]
    # extend only
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None     # int32 [B]
    extend_prefix_lens: Optional[torch.Tensor] = None  # int32 [B]
    extend_start_loc: Optional[torch.Tensor] = None    # int32 [B]
    extend_seq_lens_cpu: Optional[List[int]] = None
    extend_prefix_lens_cpu: Optional[List[int]] = None
    # pools / backend
    req_to_token_pool: object = None
    token_to_kv_pool: object = None
    attn_backend: object = None
    sampling_info: object = None
    return_logprob: bool = False
    top_logprobs_nums: Optional[List[int]] = None

    @classmethod
    def init_new(cls, batch, model_runner) -> "ForwardBatch":
        """`batch` is a ModelWorkerBatch (managers/schedule_batch.py:1227-1290)."""
        device = model_runner.device
        ret = cls(
            forward_mode=batch.forward_mode,
            batch_size=len(batch.seq_lens),
            input_ids=batch.input_ids,
            req_pool_indices=batch.req_pool_indices,
            seq_lens=batch.seq_lens,
            out_cache_loc=batch.out_cache_loc,
            seq_lens_sum=batch.seq_lens_sum,
        )
        if ret.forward_mode.is_decode():
            # decode: position = seq_len - 1 (forward_batch_info.py:299-302, clamp_position)
            ret.positions = torch.clamp(ret.seq_lens - 1, min=0).to(torch.int64)
        elif ret.forward_mode.is_extend():
            # pinned staging: a pageable copy would wait for the prefill batch that is still running (pipelined loop)
            from semi_pd_amd.managers.schedule_batch import host_list_to_device
            ret.extend_seq_lens = host_list_to_device(batch.extend_seq_lens, torch.int32, device)
            ret.extend_prefix_lens = host_list_to_device(batch.extend_prefix_lens, torch.int32, device)
            ret.extend_num_tokens = batch.extend_num_tokens
            ret.positions, ret.extend_start_loc = ops.compute_position(
                ret.extend_prefix_lens, ret.extend_seq_lens, ret.extend_num_tokens)
            ret.extend_seq_lens_cpu = batch.extend_seq_lens
            ret.extend_prefix_lens_cpu = batch.extend_prefix_lens
        ret.req_to_token_pool = model_runner.req_to_token_pool
        ret.token_to_kv_pool = model_runner.token_to_kv_pool
        ret.attn_backend = model_runner.attn_backend
        ret.sampling_info = getattr(batch, "sampling_info", None)
        ret.return_logprob = bool(getattr(batch, "return_logprob", False))
        ret.top_logprobs_nums = getattr(batch, "top_logprobs_nums", None)
        return ret
