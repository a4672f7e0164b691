"""Req / ScheduleBatch / ModelWorkerBatch / PrefillAdder / ChunkCache for the Semi-PD schedulers.

Reference: managers/schedule_batch.py (Req :234-560, ScheduleBatch.prepare_for_extend :796-990 with
the prefill-instance branch :923-937, prepare_for_decode :1144-1207, retract_decode :1034-1119,
filter_batch / merge_batch :1209-1290, get_model_worker_batch :1292-1350),
managers/schedule_policy.py:272-500 (PrefillAdder), mem_cache/chunk_cache.py (radix cache is
force-disabled in Semi-PD, server_args.py:325-331).
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import List, Optional

import torch

from semi_pd_amd.managers.io_struct import SamplingParams
from semi_pd_amd.sampling_batch_info import SamplingBatchInfo
from semi_pd_amd.model_executor.forward_batch_info import ForwardMode

CLIP_MAX_NEW_TOKENS_ESTIMATION = 4096  # schedule_policy.py:36-40


class Req:
    def __init__(self, rid: str, origin_input_ids: List[int], sampling_params: SamplingParams,
                 eos_token_ids: Optional[set] = None, is_retracted: bool = False, return_logprob: bool = False,
                 top_logprobs_num: int = 0):
        self.rid = rid
        self.return_logprob = return_logprob
        self.top_logprobs_num = int(top_logprobs_num)
        self.output_token_logprobs: List[float] = []          # one per output token
        self.output_top_logprobs: List[List[tuple]] = []      # one list of (logprob, token id) per output token
        self.origin_input_ids = list(origin_input_ids)
        self.output_ids: List[int] = []
        self.fill_ids: List[int] = []
        self.sampling_params = sampling_params
        self.eos_token_ids = eos_token_ids or set()
        self.req_pool_idx: Optional[int] = None
        self.prefix_indices = torch.empty(0, dtype=torch.int64)
        # host mirror of this request's row of req_to_token, kept by the side that allocates (the decode instance,
        # or the unified scheduler): frees never read the device table
        self.kv_slots: List[int] = []
        self.extend_input_len = 0
        self.is_chunked = 0
        self.is_retracted = is_retracted
        self.retracted_output_len = 0  # prefill instance: the last k prompt tokens were generated before a retraction
        self.finished_reason: Optional[str] = None
        self.to_abort = False  # set by AbortReq; turns into finished_reason "abort" at the next check
        self.send_token_offset = 0
        self.queue_time = 0.0

    def finished(self) -> bool:
        return self.finished_reason is not None

    def init_next_round_input(self, tree_cache=None):
        """schedule_batch.py:420-440 without a radix tree: everything not yet cached is input."""
        self.fill_ids = self.origin_input_ids + self.output_ids
        self.extend_input_len = len(self.fill_ids) - len(self.prefix_indices)

    def check_finished(self):
        """schedule_batch.py:470-520 (length and EOS; stop strings need a tokenizer -> next rows)."""
        if self.finished():
            return
        if self.to_abort:
            self.finished_reason = "abort"
            return
        if len(self.output_ids) >= self.sampling_params.max_new_tokens:
            self.finished_reason = "length"
            return
        last = self.output_ids[-1]
        if not self.sampling_params.ignore_eos:
            if last in self.eos_token_ids or (self.sampling_params.stop_token_ids and
                                              last in self.sampling_params.stop_token_ids):
                self.finished_reason = "stop"

    def reset_for_retract(self):
        self.prefix_indices = torch.empty(0, dtype=torch.int64)
        self.kv_slots = []
        self.extend_input_len = 0
        self.is_retracted = True
        self.req_pool_idx = None
        self.is_chunked = 0


class ChunkCache:
    """mem_cache/chunk_cache.py: no prefix sharing; keeps the KV indices of an unfinished (chunked)
    request and frees everything when a request finishes."""

    def __init__(self, req_to_token_pool, token_to_kv_pool_allocator):
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator

    def cache_finished_req(self, req: Req):
        # KV exists for every token but the last sampled one; a slot taken for a step that runs beyond the end of
        # the request (overlapped decode loop) stays in req.kv_slots and is released when that step is processed
        n = len(req.origin_input_ids) + len(req.output_ids) - 1
        self.req_to_token_pool.free(req.req_pool_idx)
        self.token_to_kv_pool_allocator.free(req.kv_slots[:n])
        req.kv_slots = req.kv_slots[n:]

    def cache_unfinished_req(self, req: Req):
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(req.fill_ids)]
        req.prefix_indices = kv_indices.to(torch.int64).clone()

    def evictable_size(self):
        return 0


@dataclass
class ModelWorkerBatch:
    forward_mode: ForwardMode
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    out_cache_loc: torch.Tensor
    seq_lens_sum: int
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[List[int]] = None
    extend_prefix_lens: Optional[List[int]] = None
    sampling_info: Optional[SamplingBatchInfo] = None
    return_logprob: bool = False
    top_logprobs_nums: Optional[List[int]] = None


def host_list_to_device(values, dtype, device) -> torch.Tensor:
    """A small host list as a device tensor WITHOUT synchronising the stream: a copy from pageable memory waits
    for everything queued before it (a whole decode step in the overlapped loop); staging through pinned memory
    (cached by torch's host allocator, which releases a block only after the copy has run) does not."""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


class ScheduleBatch:
    def __init__(self, reqs: List[Req], req_to_token_pool, token_to_kv_pool_allocator, tree_cache, device):
        self.reqs = reqs
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.tree_cache = tree_cache
        self.device = device
        self.forward_mode: Optional[ForwardMode] = None
        self.batch_is_full = False
        self.input_ids = self.req_pool_indices = self.seq_lens = self.out_cache_loc = None
        self.output_ids: Optional[torch.Tensor] = None
        self.seq_lens_sum = 0
        self.seq_lens_cpu: List[int] = []  # host mirror of seq_lens: bookkeeping without device round trips
        self.prefix_lens = self.extend_lens = None
        self.extend_num_tokens = 0
        self.decoding_reqs = None

    @classmethod
    def init_new(cls, reqs, req_to_token_pool, token_to_kv_pool_allocator, tree_cache, device):
        return cls(reqs, req_to_token_pool, token_to_kv_pool_allocator, tree_cache, device)

    def batch_size(self):
        return len(self.reqs)

    def is_empty(self):
        return len(self.reqs) == 0

    # ---------------------------------------------------------------------------- allocation
    def alloc_req_slots(self, num_reqs: int):
        idx = self.req_to_token_pool.alloc(num_reqs)
        if idx is None:
            raise RuntimeError("Out of memory. Please set a smaller number for `--max-running-requests`.")
        return idx

    def alloc_token_slots(self, num_tokens: int) -> torch.Tensor:
        out = self.token_to_kv_pool_allocator.alloc(num_tokens)
        if out is None:
            raise RuntimeError(f"Out of memory. Try to lower your batch size.\nTry to allocate {num_tokens} "
                               f"tokens.\nAvailable tokens: {self.token_to_kv_pool_allocator.available_size()}\n")
        return out

    # ---------------------------------------------------------------------------- extend
    def prepare_for_extend(self, pre_allocated_req_pool_indices: Optional[List[int]] = None,
                           pre_allocated_slots: Optional[List[List[int]]] = None):
        self.forward_mode = ForwardMode.EXTEND
        bs = len(self.reqs)
        if pre_allocated_req_pool_indices is None:
            req_pool_indices = self.alloc_req_slots(bs)
        else:
            assert bs == len(pre_allocated_req_pool_indices)
            req_pool_indices = list(pre_allocated_req_pool_indices)
        reqs = self.reqs
        input_ids = [r.fill_ids[len(r.prefix_indices):] for r in reqs]
        extend_num_tokens = sum(len(ids) for ids in input_ids)
        seq_lens = [len(r.fill_ids) for r in reqs]
        prefix_lens = [len(r.prefix_indices) for r in reqs]
        extend_lens = [r.extend_input_len for r in reqs]
        dev = self.device
        # (pinned staging: a pageable host -> device copy would wait for everything queued on the stream)
        self.req_pool_indices = host_list_to_device(req_pool_indices, torch.int64, dev)
        self.input_ids = host_list_to_device(sum(input_ids, []), torch.int64, dev)
        self.seq_lens = host_list_to_device(seq_lens, torch.int64, dev)
        table = self.req_to_token_pool.req_to_token
        for r, idx in zip(reqs, req_pool_indices):
            r.req_pool_idx = idx
        if pre_allocated_req_pool_indices is None:
            # decode instance: the only allocator.  Writes the *shared* table.
            out_cpu = self.alloc_token_slots(extend_num_tokens)
            out_list = out_cpu.tolist()
            out_cache_loc = host_list_to_device(out_list, torch.int64, dev)
            pt = 0
            loc32 = out_cache_loc.to(torch.int32)
            for i, r in enumerate(reqs):
                r.kv_slots = r.kv_slots[: prefix_lens[i]] + out_list[pt: pt + extend_lens[i]]
                if prefix_lens[i]:
                    table[req_pool_indices[i], : prefix_lens[i]] = r.prefix_indices.to(dev, torch.int32)
                table[req_pool_indices[i], prefix_lens[i]: seq_lens[i]] = loc32[pt: pt + extend_lens[i]]
                pt += extend_lens[i]
        else:
            # prefill instance: the slots the decode instance allocated — from its reply when it carries them, else
            # from the shared table (schedule_batch.py:923-937)
            if pre_allocated_slots is not None:
                assert [len(x) for x in pre_allocated_slots] == extend_lens
                out_cache_loc = host_list_to_device(sum(pre_allocated_slots, []), torch.int64, dev)
            else:
                parts = [table[idx, pre:seq] for idx, pre, seq in zip(req_pool_indices, prefix_lens, seq_lens)]
                out_cache_loc = torch.cat(parts).to(dev, dtype=torch.int64)
        self.out_cache_loc = out_cache_loc
        self.seq_lens_sum = sum(seq_lens)
        self.seq_lens_cpu = list(seq_lens)
        self.extend_num_tokens = extend_num_tokens
        self.prefix_lens = prefix_lens
        self.extend_lens = extend_lens

    # ---------------------------------------------------------------------------- decode
    def check_decode_mem(self, buf_multiplier: int = 1) -> bool:
        return self.token_to_kv_pool_allocator.available_size() >= len(self.reqs) * buf_multiplier

    def retract_decode(self, force: int = 0):
        """schedule_batch.py:1034-1119: drop requests (fewest output tokens first) until one decode
        step fits; `force` retracts at least that many (SGLANG_TEST_RETRACT)."""
        order = sorted(range(len(self.reqs)),
                       key=lambda i: (len(self.reqs[i].output_ids), -len(self.reqs[i].origin_input_ids)),
                       reverse=True)
        retracted = []
        while (self.token_to_kv_pool_allocator.available_size() < len(order) or len(retracted) < force) \
                and len(order) > 1:
            i = order.pop()
            req = self.reqs[i]
            retracted.append(req)
            n = len(req.origin_input_ids) + len(req.output_ids) - 1
            self.token_to_kv_pool_allocator.free(req.kv_slots[:n])
            self.req_to_token_pool.free(req.req_pool_idx)
            req.reset_for_retract()
        self.filter_batch(keep_indices=sorted(order))
        total_decoded = sum(len(r.output_ids) for r in self.reqs)
        total_max = sum(r.sampling_params.max_new_tokens for r in self.reqs)
        new_ratio = min(1.0, (total_decoded + 5 * len(self.reqs)) / max(total_max, 1))
        return retracted, new_ratio

    def prepare_for_decode(self):
        self.forward_mode = ForwardMode.DECODE
        bs = len(self.reqs)
        self.input_ids = self.output_ids.to(torch.int64)
        self.output_ids = None
        new_slots = self.alloc_token_slots(bs).tolist()
        for r, slot in zip(self.reqs, new_slots):
            r.kv_slots.append(slot)
        self.out_cache_loc = host_list_to_device(new_slots, torch.int64, self.device)
        # req_to_token[req, seq_len] = new slot; then seq_len += 1 (schedule_batch.py:1190-1205)
        self.req_to_token_pool.req_to_token[self.req_pool_indices, self.seq_lens] = self.out_cache_loc.to(torch.int32)
        self.seq_lens = self.seq_lens + 1
        self.seq_lens_sum += bs
        self.seq_lens_cpu = [x + 1 for x in self.seq_lens_cpu]

    # ---------------------------------------------------------------------------- bookkeeping
    def filter_batch(self, chunked_req_to_exclude: Optional[Req] = None, keep_indices: Optional[List[int]] = None):
        if keep_indices is None:
            keep_indices = [i for i in range(len(self.reqs))
                            if not self.reqs[i].finished() and self.reqs[i] is not chunked_req_to_exclude]
        if len(keep_indices) == len(self.reqs):
            return
        if not keep_indices:
            self.reqs = []
            self.seq_lens_cpu = []
            return
        self.reqs = [self.reqs[i] for i in keep_indices]
        idx = host_list_to_device(keep_indices, torch.int64, self.device)
        self.req_pool_indices = self.req_pool_indices[idx]
        self.seq_lens = self.seq_lens[idx]
        self.out_cache_loc = None
        self.seq_lens_cpu = [self.seq_lens_cpu[i] for i in keep_indices]
        self.seq_lens_sum = sum(self.seq_lens_cpu)  # (no .item(): the decode loop must not wait for the GPU here)
        if self.output_ids is not None:
            self.output_ids = self.output_ids[idx]

    def merge_batch(self, other: "ScheduleBatch"):
        self.req_pool_indices = torch.concat([self.req_pool_indices, other.req_pool_indices])
        self.seq_lens = torch.concat([self.seq_lens, other.seq_lens])
        self.out_cache_loc = None
        self.seq_lens_sum += other.seq_lens_sum
        self.seq_lens_cpu = self.seq_lens_cpu + other.seq_lens_cpu
        if self.output_ids is not None and other.output_ids is not None:
            self.output_ids = torch.concat([self.output_ids.to(torch.int64), other.output_ids.to(torch.int64)])
        self.reqs.extend(other.reqs)

    def get_model_worker_batch(self) -> ModelWorkerBatch:
        ext = self.forward_mode.is_extend()
        return ModelWorkerBatch(
            forward_mode=self.forward_mode, input_ids=self.input_ids, req_pool_indices=self.req_pool_indices,
            seq_lens=self.seq_lens, out_cache_loc=self.out_cache_loc, seq_lens_sum=self.seq_lens_sum,
            extend_num_tokens=self.extend_num_tokens if ext else None,
            extend_seq_lens=self.extend_lens if ext else None,
            extend_prefix_lens=self.prefix_lens if ext else None,
            sampling_info=SamplingBatchInfo.from_reqs(self.reqs, getattr(self, "vocab_size", 0), self.device),
            return_logprob=any(r.return_logprob for r in self.reqs),
            top_logprobs_nums=[r.top_logprobs_num if r.return_logprob else 0 for r in self.reqs])


class AddReqResult(Enum):
    CONTINUE = auto()
    NO_TOKEN = auto()
    OTHER = auto()


class PrefillAdder:
    """Token-budget admission of waiting requests into one prefill batch
    (managers/schedule_policy.py:272-500, no-radix path)."""

    def __init__(self, tree_cache, token_to_kv_pool_allocator, running_batch: Optional[ScheduleBatch],
                 new_token_ratio: float, rem_input_tokens: int, rem_chunk_tokens: Optional[int]):
        self.rem_total_tokens = token_to_kv_pool_allocator.available_size() + tree_cache.evictable_size()
        self.rem_input_tokens = rem_input_tokens
        self.rem_chunk_tokens = rem_chunk_tokens
        self.can_run_list: List[Req] = []
        self.new_chunked_req: Optional[Req] = None
        self.log_input_tokens = 0
        if running_batch is not None:
            self.rem_total_tokens -= sum(
                min(r.sampling_params.max_new_tokens - len(r.output_ids), CLIP_MAX_NEW_TOKENS_ESTIMATION)
                * new_token_ratio for r in running_batch.reqs)

    def budget_state(self):
        if self.rem_total_tokens <= 0:
            return AddReqResult.NO_TOKEN
        if self.rem_input_tokens <= 0 or (self.rem_chunk_tokens is not None and self.rem_chunk_tokens <= 0):
            return AddReqResult.OTHER
        return AddReqResult.CONTINUE

    def _prefill_one_req(self, extend_input_len: int, max_new_tokens: int):
        self.rem_total_tokens -= extend_input_len + max_new_tokens
        self.rem_input_tokens -= extend_input_len
        if self.rem_chunk_tokens is not None:
            self.rem_chunk_tokens -= extend_input_len
        self.log_input_tokens += extend_input_len

    def add_chunked_req(self, req: Req) -> Optional[Req]:
        truncated = req.extend_input_len > self.rem_chunk_tokens
        req.extend_input_len = min(req.extend_input_len, self.rem_chunk_tokens)
        req.fill_ids = req.fill_ids[: len(req.prefix_indices) + req.extend_input_len]
        self.can_run_list.append(req)
        self._prefill_one_req(req.extend_input_len,
                              min(req.sampling_params.max_new_tokens, CLIP_MAX_NEW_TOKENS_ESTIMATION)
                              if not truncated else 0)
        return req if truncated else None

    def add_one_req(self, req: Req) -> AddReqResult:
        total_tokens = req.extend_input_len + min(req.sampling_params.max_new_tokens,
                                                  CLIP_MAX_NEW_TOKENS_ESTIMATION)
        input_tokens = req.extend_input_len
        if total_tokens >= self.rem_total_tokens:
            return AddReqResult.NO_TOKEN
        if input_tokens > self.rem_input_tokens and len(self.can_run_list) != 0:
            return AddReqResult.OTHER
        if self.rem_chunk_tokens is None or input_tokens <= self.rem_chunk_tokens:
            self.can_run_list.append(req)
            self._prefill_one_req(input_tokens, min(req.sampling_params.max_new_tokens,
                                                    CLIP_MAX_NEW_TOKENS_ESTIMATION))
        else:
            trunc_len = self.rem_chunk_tokens
            if trunc_len <= 0:
                return AddReqResult.OTHER
            req.extend_input_len = trunc_len
            req.fill_ids = req.fill_ids[: len(req.prefix_indices) + trunc_len]
            self.can_run_list.append(req)
            self.new_chunked_req = req
            self._prefill_one_req(trunc_len, 0)
        return self.budget_state()
