"""Scheduler base + the unified (single-process) scheduler.

Reference: managers/scheduler.py — recv_requests :599-660, process_input_requests :662-700,
handle_generate_request :702-830, get_next_batch_to_run :1031-1108, get_new_batch_prefill :1110-1247,
update_running_batch :1249-1290, run_batch :1292-1340, process_batch_result_prefill/decode
:1342-1500, stream_output :1502-1650, watchdog :1455-1484; managers/tp_worker.py:182-198.

The unified scheduler exists to pin the Semi-PD protocol: the reference has no Semi-PD test, so
the invariant "Semi-PD greedy tokens == unified-engine tokens" is the oracle for a15.
"""
from __future__ import annotations

import logging
import time
from typing import Dict, List, Optional

import torch

from semi_pd_amd.distributed import broadcast_pyobj, get_tp_cpu_group
from semi_pd_amd.managers.io_struct import (AbortReq, BatchTokenIDOut, FlushCacheReq, ShutdownReq, StatsReq,
                                            TokenizedGenerateReqInput)
from semi_pd_amd.managers.schedule_batch import (AddReqResult, ChunkCache, PrefillAdder, Req, ScheduleBatch)
from semi_pd_amd.managers.transport import NOTHING
from semi_pd_amd.model_executor.forward_batch_info import ForwardBatch
from semi_pd_amd.semi_pd.utils import InstanceRole

logger = logging.getLogger(__name__)


class TpModelWorker:
    """managers/tp_worker.py:182-198: ModelWorkerBatch -> (logits_output, next_token_ids)."""

    def __init__(self, model_runner):
        self.model_runner = model_runner
        self.device = model_runner.device

    def forward_batch_generation(self, model_worker_batch):
        forward_batch = ForwardBatch.init_new(model_worker_batch, self.model_runner)
        logits_output = self.model_runner.forward(forward_batch)
        next_token_ids = self.model_runner.sample(logits_output, forward_batch)
        return logits_output, next_token_ids


class SchedulerBase:
    """State and helpers shared by the unified, prefill and decode schedulers."""

    def __init__(self, server_args, model_runner, tp_rank: int, recv_socket, send_to_detokenizer,
                 role: InstanceRole):
        self.server_args = server_args
        self.role = role
        self.tp_rank = tp_rank
        self.tp_size = server_args.tp_size
        self.tp_cpu_group = get_tp_cpu_group()
        self.model_runner = model_runner
        self.tp_worker = TpModelWorker(model_runner)
        self.device = model_runner.device
        self.recv_from_tokenizer = recv_socket
        self.send_to_detokenizer = send_to_detokenizer
        self.req_to_token_pool = model_runner.req_to_token_pool
        self.token_to_kv_pool_allocator = model_runner.token_to_kv_pool_allocator
        self.tree_cache = ChunkCache(self.req_to_token_pool, self.token_to_kv_pool_allocator)
        self.max_total_num_tokens = model_runner.max_total_num_tokens
        self.max_running_requests = server_args.max_running_requests
        self.max_prefill_tokens = server_args.max_prefill_tokens
        self.chunked_prefill_size = server_args.chunked_prefill_size
        self.max_req_input_len = server_args.context_length - 1
        self.eos_token_ids = set(server_args.eos_token_ids or [])
        self.waiting_queue: List[Req] = []
        self.running_batch = ScheduleBatch([], self.req_to_token_pool, self.token_to_kv_pool_allocator,
                                           self.tree_cache, self.device)
        self.chunked_req: Optional[Req] = None
        # new-token-ratio estimator (scheduler.py:330-345)
        self.init_new_token_ratio = min(0.7 * server_args.schedule_conservativeness, 1.0)
        self.min_new_token_ratio = min(self.init_new_token_ratio * 0.14, 1.0)
        self.new_token_ratio_decay = (self.init_new_token_ratio - self.min_new_token_ratio) / 600
        self.new_token_ratio = self.init_new_token_ratio
        self.forward_ct = 0
        self.last_progress = time.monotonic()
        self.stats: Dict[str, float] = {"prefill_batches": 0, "prefill_tokens": 0, "decode_steps": 0,
                                        "decode_tokens": 0}
        self._shutdown = False
        self._deferred_out: List[BatchTokenIDOut] = []
        self.defer_decode_stream = False  # the Semi-PD decode loop turns this on

    # ---------------------------------------------------------------------------- input
    def recv_requests(self) -> list:
        """zmq NOBLOCK drain on rank 0 + pickle broadcast to TP peers (scheduler.py:599-660)."""
        recv_reqs = []
        if self.tp_rank == 0 and self.recv_from_tokenizer is not None:
            while True:
                obj = self.recv_from_tokenizer.recv_pyobj_nowait()
                if obj is NOTHING:
                    break
                recv_reqs.append(obj)
        if self.tp_size > 1:
            recv_reqs = broadcast_pyobj(recv_reqs, self.tp_rank, self.tp_cpu_group, src=0)
        return recv_reqs

    def process_input_requests(self, recv_reqs: list):
        for r in recv_reqs:
            self.dispatch(r)

    def dispatch(self, recv_req):
        if isinstance(recv_req, TokenizedGenerateReqInput):
            self.handle_generate_request(recv_req)
        elif isinstance(recv_req, FlushCacheReq):
            pass
        elif isinstance(recv_req, AbortReq):
            self.abort_request(recv_req)
        elif isinstance(recv_req, ShutdownReq):
            self._shutdown = True
        elif isinstance(recv_req, StatsReq):
            self.handle_stats(recv_req)
        else:
            raise ValueError(f"Invalid request: {recv_req}")

    def handle_generate_request(self, recv_req: TokenizedGenerateReqInput):
        req = Req(recv_req.rid, recv_req.input_ids, recv_req.sampling_params, self.eos_token_ids,
                  is_retracted=recv_req.is_retracted, return_logprob=recv_req.return_logprob,
                  top_logprobs_num=recv_req.top_logprobs_num)
        req.retracted_output_len = getattr(recv_req, "retracted_output_len", 0)
        if len(req.origin_input_ids) > self.max_req_input_len:
            # validate_input_length (scheduler.py:760-775): truncate
            req.origin_input_ids = req.origin_input_ids[: self.max_req_input_len]
        req.sampling_params.max_new_tokens = min(
            req.sampling_params.max_new_tokens,
            self.server_args.context_length - len(req.origin_input_ids) - 1)
        req.queue_time = time.monotonic()
        self.add_to_waiting_queue(req)

    def add_to_waiting_queue(self, req: Req):
        self.waiting_queue.append(req)

    def abort_request(self, recv_req: AbortReq):
        """scheduler.py:1565-1584: drop a queued request, or mark a running one so that it finishes (and
        frees its KV slots) at the next step.  Requests the prefill instance is working on right now are
        marked too and end right after they are merged."""
        for i, req in enumerate(self.waiting_queue):
            if req.rid == recv_req.rid:
                self.waiting_queue.pop(i)
                return
        pending = [r for b in getattr(self, "scheduled_prefill_batches", []) for r in b.reqs]
        chunked = [self.chunked_req] if self.chunked_req is not None else []
        for req in list(self.running_batch.reqs) + pending + chunked:
            if req.rid == recv_req.rid and not req.finished():
                req.to_abort = True
                return

    def handle_stats(self, recv_req: StatsReq):
        if self.send_to_detokenizer is not None and self.tp_rank == 0:
            out = dict(self.stats)
            out["role"] = self.role.name
            out["available_kv_slots"] = int(self.token_to_kv_pool_allocator.available_size())
            out["num_running_reqs"] = len(self.running_batch.reqs)
            out["num_waiting_reqs"] = len(self.waiting_queue)
            extra = getattr(self.model_runner, "kernel_timing", None)
            if extra is not None:
                out["kernel_timing"] = extra.summary()
                if recv_req.reset:
                    extra.reset()
            from semi_pd_amd.distributed import OVERLAP_STATS
            out["all_reduce_overlap"] = dict(OVERLAP_STATS)
            pacer = getattr(self.model_runner, "step_pacer", None)
            if pacer is not None:
                # the prefill instance's decode-step deadline (semi_pd/step_pacer.py): layer gates passed, holds, time held
                out["step_gate"] = pacer.stats()
                if recv_req.reset:
                    pacer.reset_stats()
            self.send_to_detokenizer.send_pyobj(("stats", out))
        if recv_req.reset:
            for k in self.stats:
                self.stats[k] = 0

    # ---------------------------------------------------------------------------- run
    def run_batch(self, batch: ScheduleBatch):
        self.forward_ct += 1
        mwb = batch.get_model_worker_batch()
        logits_output, next_token_ids = self.tp_worker.forward_batch_generation(mwb)
        return logits_output, next_token_ids

    # ---------------------------------------------------------------------------- results
    @staticmethod
    def extract_logprobs(logits_output) -> Optional[dict]:
        """Host copy of what the sampler attached for requests with return_logprob (one sync)."""
        lp = getattr(logits_output, "next_token_logprobs", None)
        if lp is None:
            return None
        return {"token": lp.tolist(), "top_val": logits_output.next_token_top_logprobs_val,
                "top_idx": logits_output.next_token_top_logprobs_idx}

    @staticmethod
    def _record_logprob(req: Req, i: int, logprobs: Optional[dict]):
        if not req.return_logprob or logprobs is None:
            return
        req.output_token_logprobs.append(float(logprobs["token"][i]))
        if req.top_logprobs_num > 0 and logprobs.get("top_val") is not None:
            req.output_top_logprobs.append(list(zip(logprobs["top_val"][i], logprobs["top_idx"][i])))
        else:
            req.output_top_logprobs.append([])

    def process_batch_result_prefill(self, batch: ScheduleBatch, next_token_ids: List[int],
                                     logprobs: Optional[dict] = None):
        """scheduler.py:1342-1420: append the first token, finish / stream, keep chunked reqs pending."""
        out_reqs = []
        for i, (req, tok) in enumerate(zip(batch.reqs, next_token_ids)):
            if req.is_chunked <= 0:
                req.output_ids.append(int(tok))
                self._record_logprob(req, i, logprobs)
                req.check_finished()
                if req.finished():
                    self.tree_cache.cache_finished_req(req)
                out_reqs.append(req)
            else:
                req.is_chunked -= 1  # an intermediate chunk: its sampled token is discarded
        self.stats["prefill_batches"] += 1
        self.stats["prefill_tokens"] += batch.extend_num_tokens
        self.stream_output(out_reqs)
        self.last_progress = time.monotonic()

    def process_batch_result_decode(self, batch: ScheduleBatch, next_token_ids: List[int],
                                    logprobs: Optional[dict] = None):
        self.token_to_kv_pool_allocator.free_group_begin()
        for i, (req, tok) in enumerate(zip(batch.reqs, next_token_ids)):
            req.output_ids.append(int(tok))
            self._record_logprob(req, i, logprobs)
            req.check_finished()
            if req.finished():
                self.tree_cache.cache_finished_req(req)
        self.token_to_kv_pool_allocator.free_group_end()
        self.stats["decode_steps"] += 1
        self.stats["decode_tokens"] += len(batch.reqs)
        self.stream_output(batch.reqs, defer=self.defer_decode_stream)
        self.last_progress = time.monotonic()

    def stream_output(self, reqs: List[Req], defer: bool = False):
        """Send the tokens produced since the last message.  With defer=True the message is built now
        but pickled and sent by flush_stream_output(), which the decode loop calls right after it has
        launched the NEXT step, so that the host cost hides behind GPU work (the role of the overlap
        thread of the reference, managers/tp_worker_overlap_thread.py, for the output path)."""
        if self.tp_rank != 0 or self.send_to_detokenizer is None or not reqs:
            return
        now = time.time()
        rids, fins, outs, lps, tops = [], [], [], [], []
        for r in reqs:
            new = r.output_ids[r.send_token_offset:]
            if not new and not r.finished():
                continue
            lps.append(r.output_token_logprobs[r.send_token_offset:] if r.return_logprob else None)
            tops.append(r.output_top_logprobs[r.send_token_offset:] if r.return_logprob else None)
            r.send_token_offset = len(r.output_ids)
            rids.append(r.rid)
            fins.append(r.finished_reason)
            outs.append(new)
        if rids:
            with_lp = any(x is not None for x in lps)
            msg = BatchTokenIDOut(rids, fins, outs, [now] * len(rids), lps if with_lp else None,
                                  tops if with_lp else None)
            if defer:
                self._deferred_out.append(msg)
            else:
                self.flush_stream_output()  # keep per-request order
                self.send_to_detokenizer.send_pyobj(msg)

    def flush_stream_output(self):
        if self._deferred_out:
            pending, self._deferred_out = self._deferred_out, []
            for msg in pending:
                self.send_to_detokenizer.send_pyobj(msg)

    # ---------------------------------------------------------------------------- decode helpers
    def update_running_batch(self, batch: ScheduleBatch) -> ScheduleBatch:
        """scheduler.py:1249-1290 (the decode instance overrides this to re-route retracted requests)."""
        initial_bs = batch.batch_size()
        batch.filter_batch()
        if batch.is_empty():
            batch.batch_is_full = False
            return batch
        force = self.forced_retractions(batch)
        if not batch.check_decode_mem() or force:
            retracted, self.new_token_ratio = batch.retract_decode(force=force)
            logger.info("Decode out of memory happened. #retracted_reqs: %d", len(retracted))
            self.on_retract(retracted)
        else:
            self.new_token_ratio = max(self.new_token_ratio - self.new_token_ratio_decay, self.min_new_token_ratio)
        if batch.batch_size() < initial_bs:
            batch.batch_is_full = False
        batch.prepare_for_decode()
        return batch

    def forced_retractions(self, batch: ScheduleBatch) -> int:
        return 0

    def on_retract(self, retracted: List[Req]):
        for r in retracted:
            self.waiting_queue.insert(0, r)

    def build_prefill_adder(self) -> PrefillAdder:
        return PrefillAdder(self.tree_cache, self.token_to_kv_pool_allocator, self.running_batch,
                            self.new_token_ratio, self.max_prefill_tokens, self.chunked_prefill_size)

    def check_watchdog(self):
        """scheduler.py:1455-1484: no forward progress while work is pending -> fail fast."""
        busy = bool(self.waiting_queue) or not self.running_batch.is_empty()
        if busy and time.monotonic() - self.last_progress > self.server_args.watchdog_timeout:
            raise RuntimeError(f"{self.role.name} scheduler watchdog timeout "
                               f"({self.server_args.watchdog_timeout}s without forward progress)")

    def idle_sleep(self):
        time.sleep(0.0002)


class Scheduler(SchedulerBase):
    """Unified engine: prefill has priority over decode in one process (scheduler.py:1031-1108)."""

    def __init__(self, server_args, model_runner, tp_rank, recv_socket, send_to_detokenizer):
        super().__init__(server_args, model_runner, tp_rank, recv_socket, send_to_detokenizer, InstanceRole.OTHER)
        self.last_batch: Optional[ScheduleBatch] = None

    def get_new_batch_prefill(self) -> Optional[ScheduleBatch]:
        if (self.running_batch.batch_is_full or len(self.waiting_queue) == 0) and self.chunked_req is None:
            return None
        running_bs = len(self.running_batch.reqs)
        if running_bs >= self.max_running_requests:
            self.running_batch.batch_is_full = True
            return None
        adder = self.build_prefill_adder()
        if self.chunked_req is not None:
            self.chunked_req.init_next_round_input()
            self.chunked_req = adder.add_chunked_req(self.chunked_req)
        for req in self.waiting_queue:
            if running_bs + len(adder.can_run_list) >= self.max_running_requests:
                self.running_batch.batch_is_full = True
                break
            req.init_next_round_input()
            res = adder.add_one_req(req)
            if res != AddReqResult.CONTINUE:
                if res == AddReqResult.NO_TOKEN:
                    self.running_batch.batch_is_full = True
                break
        can_run_list = adder.can_run_list
        if len(can_run_list) == 0:
            return None
        chosen = set(id(x) for x in can_run_list)
        self.waiting_queue = [x for x in self.waiting_queue if id(x) not in chosen]
        if adder.new_chunked_req is not None:
            assert self.chunked_req is None
            self.chunked_req = adder.new_chunked_req
        if self.chunked_req:
            self.chunked_req.is_chunked += 1
        new_batch = ScheduleBatch.init_new(can_run_list, self.req_to_token_pool, self.token_to_kv_pool_allocator,
                                           self.tree_cache, self.device)
        new_batch.prepare_for_extend()
        return new_batch

    def get_next_batch_to_run(self) -> Optional[ScheduleBatch]:
        if self.last_batch is not None and self.last_batch.forward_mode.is_extend():
            if self.chunked_req:
                # move the chunked request out of the batch; keep its KV (scheduler.py:1040-1050)
                self.last_batch.filter_batch(chunked_req_to_exclude=self.chunked_req)
                self.tree_cache.cache_unfinished_req(self.chunked_req)
                self.req_to_token_pool.free(self.chunked_req.req_pool_idx)
                self.running_batch.batch_is_full = False
            else:
                last_bs = self.last_batch.batch_size()
                self.last_batch.filter_batch()
                if self.last_batch.batch_size() < last_bs:
                    # requests that ended at their first token gave their slots back (scheduler.py:1043-1047)
                    self.running_batch.batch_is_full = False
            if not self.last_batch.is_empty():
                if self.running_batch.is_empty():
                    self.running_batch = self.last_batch
                else:
                    self.running_batch.merge_batch(self.last_batch)
        if self.running_batch.is_empty():
            self.running_batch.batch_is_full = False  # nothing runs: every slot is free again
        new_batch = self.get_new_batch_prefill()
        if new_batch is not None:
            return new_batch
        if not self.running_batch.is_empty():
            self.running_batch = self.update_running_batch(self.running_batch)
            return self.running_batch if not self.running_batch.is_empty() else None
        return None

    def step(self) -> bool:
        """One scheduler iteration; returns False when idle."""
        self.process_input_requests(self.recv_requests())
        batch = self.get_next_batch_to_run()
        if batch is None:
            self.last_batch = None
            return False
        logits_output, next_token_ids = self.run_batch(batch)
        ids = next_token_ids.tolist()
        logprobs = self.extract_logprobs(logits_output)
        if batch.forward_mode.is_extend():
            batch.output_ids = next_token_ids
            self.process_batch_result_prefill(batch, ids, logprobs)
        else:
            batch.output_ids = next_token_ids
            self.process_batch_result_decode(batch, ids, logprobs)
        self.last_batch = batch
        return True

    def event_loop_normal(self):
        while not self._shutdown:
            if not self.step():
                self.check_watchdog()
                self.idle_sleep()
