"""SemiPDDecodeScheduler: the decode instance owns every allocation.  It answers the prefill
instance's "which of these rids may I prefill?" by running the real PrefillAdder, allocating the
request slot and the KV slots and writing them into the shared req_to_token table; later it merges
the prefilled batch into its running decode batch.  Reference:
managers/semi_pd_decode_scheduler.py:44-377."""
from __future__ import annotations

import logging
import os
import time
from typing import List, Optional

import torch

from semi_pd_amd.distributed import barrier_cpu
from semi_pd_amd.managers.io_struct import (BatchProcessPrefillResultReq, GetNextPrefillBatchInput,
                                            GetNextPrefillBatchOutput, SyntheticLoadReq, TokenizedGenerateReqInput)
from semi_pd_amd.managers.schedule_batch import AddReqResult, Req, ScheduleBatch, host_list_to_device
from semi_pd_amd.managers.scheduler import SchedulerBase
from semi_pd_amd.semi_pd import ttft_trace
from semi_pd_amd.semi_pd.utils import InstanceRole

logger = logging.getLogger(__name__)

TEST_RETRACT = os.environ.get("SGLANG_TEST_RETRACT", "0").lower() in ("1", "true")


class SemiPDDecodeScheduler(SchedulerBase):
    def __init__(self, server_args, model_runner, tp_rank, recv_socket, send_to_detokenizer, bridge_socket,
                 send_to_p_instance):
        super().__init__(server_args, model_runner, tp_rank, recv_socket, send_to_detokenizer, InstanceRole.DECODE)
        # requests handed to the prefill instance whose result has not come back yet
        self.scheduled_prefill_batches: List[ScheduleBatch] = []
        self.defer_decode_stream = True               # see step(): outputs go out behind the next launch
        # overlap schedule (managers/tp_worker_overlap_thread.py, scheduler.py event_loop_overlap; on unless
        # --disable-overlap-schedule): step k + 1 is launched before the tokens of step k are looked at
        self.enable_overlap = not getattr(server_args, "disable_overlap_schedule", False)
        self._pending = None                          # (reqs, out_cache_loc, pinned ids, event, logits_output) of step k
        self._in_wait = False                         # inside _wait_servicing (it must not nest)
        self._deferred_input: list = []               # messages a wait set aside for the loop top
        self._pinned_ids = None
        self._pinned_flip = 0
        self._synthetic_load_until = 0.0              # see SyntheticLoadReq
        self._published_busy = -1                     # what the share board last heard from this instance

        self.bridge_socket = bridge_socket            # PUSH -> P (replies to GetNextPrefillBatchInput)
        self.send_to_p_instance = send_to_p_instance  # PUSH -> P's input socket (retracted requests)

    # ---------------------------------------------------------------------------- dispatch
    def dispatch(self, recv_req):
        if isinstance(recv_req, GetNextPrefillBatchInput):
            self.get_next_prefill_batch(recv_req)
        elif isinstance(recv_req, BatchProcessPrefillResultReq):
            self.process_prefill_result(recv_req)
        elif isinstance(recv_req, SyntheticLoadReq):
            # bounded: a prefill instance that dies while tuning must not leave this one spinning
            self._synthetic_load_until = time.monotonic() + 300.0 if recv_req.on else 0.0
        else:
            super().dispatch(recv_req)

    def add_to_waiting_queue(self, req: Req):
        if req.is_retracted:
            return  # D re-queued it itself when it retracted it (semi_pd_decode_scheduler.py:137)
        self.waiting_queue.append(req)

    # ---------------------------------------------------------------------------- decode side
    def on_retract(self, retracted: List[Req]):
        """semi_pd_decode_scheduler.py:117-139: a retracted request goes back to the front of D's
        queue and is re-sent to P with its generated tokens appended to the prompt."""
        self.stats["retracted_reqs"] = self.stats.get("retracted_reqs", 0) + len(retracted)
        for req in retracted:
            message = TokenizedGenerateReqInput(
                rid=req.rid, input_text=None, input_ids=req.origin_input_ids + req.output_ids,
                sampling_params=req.sampling_params, is_retracted=True, return_logprob=req.return_logprob,
                top_logprobs_num=req.top_logprobs_num, retracted_output_len=len(req.output_ids))
            self.waiting_queue.insert(0, req)
            if self.tp_rank == 0:
                self.send_to_p_instance.send_pyobj(message)

    def forced_retractions(self, batch: ScheduleBatch) -> int:
        """SGLANG_TEST_RETRACT (semi_pd_decode_scheduler.py:42-43, 103-105): retract although memory is
        available, to exercise the D -> P re-send path."""
        return 2 if (TEST_RETRACT and batch.batch_size() > 10) else 0

    def get_next_batch_to_run(self) -> Optional[ScheduleBatch]:
        """semi_pd_decode_scheduler.py:141-153: D only ever runs decode batches."""
        if not self.running_batch.is_empty():
            self.running_batch = self.update_running_batch(self.running_batch)
            return self.running_batch if not self.running_batch.is_empty() else None
        return None

    # ---------------------------------------------------------------------------- prefill admission
    def get_new_batch_prefill(self, rids: List[str]) -> Optional[ScheduleBatch]:
        """semi_pd_decode_scheduler.py:155-308 (radix / LoRA / hierarchical-cache branches do not exist
        in Semi-PD mode)."""
        if (self.running_batch.batch_is_full or len(self.waiting_queue) == 0) and self.chunked_req is None:
            return None
        pending = sum(len(b.reqs) for b in self.scheduled_prefill_batches)
        running_bs = len(self.running_batch.reqs) + pending
        if running_bs >= self.max_running_requests:
            self.running_batch.batch_is_full = True
            return None
        adder = self.build_prefill_adder()
        for b in self.scheduled_prefill_batches:  # tokens promised to in-flight prefills are spoken for
            for r in b.reqs:
                adder.rem_total_tokens -= min(r.sampling_params.max_new_tokens, 4096) * self.new_token_ratio
        if self.chunked_req is not None:
            self.chunked_req.init_next_round_input()
            self.chunked_req = adder.add_chunked_req(self.chunked_req)
        rid_set = set(rids)
        for req in self.waiting_queue:
            if req.rid not in rid_set:
                continue
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
        self.scheduled_prefill_batches.append(new_batch)
        new_batch.decoding_reqs = None
        return new_batch

    def get_next_prefill_batch(self, recv_req: GetNextPrefillBatchInput):
        """semi_pd_decode_scheduler.py:310-337."""
        if self.chunked_req and self.scheduled_prefill_batches:
            # the previous chunk is still running in P: answer "nothing yet"
            self._reply_empty()
            return
        ttft_trace.mark("d_got_proposal", recv_req.rids)
        batch = self._admit(recv_req)
        if batch is None:
            self._reply_empty()
            return
        prefix_lens = [len(r.prefix_indices) for r in batch.reqs]
        # P reads the shared table only for PREFIX tokens (a later chunk of a long prompt, a re-sent request): only
        # then must the table be complete in HBM before the reply (the reference relies on timing).  The slots of
        # the tokens to prefill go with the reply, so the common admission needs no GPU synchronisation at all —
        # which, with a decode step always in flight (overlapped loop), would cost a whole step.
        if any(prefix_lens) and torch.device(self.device).type == "cuda":
            torch.cuda.current_stream().synchronize()
        if self.tp_rank == 0:
            self.bridge_socket.send_pyobj(GetNextPrefillBatchOutput(
                rids=[r.rid for r in batch.reqs],
                chunked_rid=(self.chunked_req.rid if self.chunked_req else None),
                req_pool_indices=[r.req_pool_idx for r in batch.reqs],
                prefix_lens=prefix_lens,
                extend_input_lens=[r.extend_input_len for r in batch.reqs],
                extend_slots=[r.kv_slots[p: p + r.extend_input_len] for r, p in zip(batch.reqs, prefix_lens)]))

    def _admit(self, recv_req: GetNextPrefillBatchInput) -> Optional[ScheduleBatch]:
        if self.chunked_req:
            self.tree_cache.cache_unfinished_req(self.chunked_req)
            self.req_to_token_pool.free(self.chunked_req.req_pool_idx)
        return self.get_new_batch_prefill(recv_req.rids)

    def _reply_empty(self):
        if self.tp_rank == 0:
            self.bridge_socket.send_pyobj(GetNextPrefillBatchOutput(
                rids=[], chunked_rid=(self.chunked_req.rid if self.chunked_req else None),
                req_pool_indices=[], prefix_lens=[], extend_input_lens=[]))

    def process_prefill_result(self, recv_req: BatchProcessPrefillResultReq):
        """semi_pd_decode_scheduler.py:339-377."""
        batch = self.scheduled_prefill_batches.pop(0)
        assert len(batch.reqs) == len(recv_req.next_token_ids)
        ttft_trace.mark("d_got_result", [r.rid for r in batch.reqs])
        if self.tp_size > 1:
            barrier_cpu()
        batch.output_ids = host_list_to_device(recv_req.next_token_ids, torch.int64, self.device)
        self.process_batch_result_prefill(batch, recv_req.next_token_ids, recv_req.next_token_logprobs)
        ttft_trace.mark("d_streamed", [r.rid for r in batch.reqs])
        n_before = len(batch.reqs)
        batch.filter_batch(chunked_req_to_exclude=self.chunked_req)
        if len(batch.reqs) < n_before or self.running_batch.is_empty():
            # requests that ended at their first token (max_new_tokens = 1, EOS) free their slots here; the
            # "full" flag is otherwise only cleared by update_running_batch, which never runs on an empty
            # running batch (the unified path resets it when the last batch shrinks, scheduler.py:1043-1047)
            self.running_batch.batch_is_full = False
        if not batch.is_empty():
            if self.running_batch.is_empty():
                self.running_batch = batch
            else:
                self.running_batch.merge_batch(batch)

    # ---------------------------------------------------------------------------- CU share
    def _share_step(self, running: int):
        """--cu-mask-mode dynamic (model_executor/cu_share.py): tell the prefill instance how many requests this step
        decodes (0: nothing in flight -- it may take every CU) and put the step on the whole chip while the prefill
        instance has no batch in flight, on the decode share otherwise."""
        share = getattr(self.model_runner, "cu_share", None)
        if share is None:
            return
        if running or running != self._published_busy:   # (a busy instance refreshes its heartbeat with every step)
            share.publish(running)
            self._published_busy = running
        if running:
            name = share.step()
            self.stats["steps_on_" + name] = self.stats.get("steps_on_" + name, 0) + 1

    def _publish_step(self, started: bool) -> None:
        """The share board's STEP_START_NS / STEP_SEQ: when the decode step in flight began on the GPU (the prefill
        instance's step pacer holds its launches while that step is overdue, semi_pd/step_pacer.py), and STEP_FAST_NS: the
        10th percentile of the recent step times -- what a step costs when nothing is in its way."""
        share = getattr(self.model_runner, "cu_share", None)
        board = getattr(share, "board", None) if share is not None else None
        if board is None:
            return
        now = time.monotonic_ns()
        prev = getattr(self, "_step_started_ns", 0)
        if prev:
            hist = self.__dict__.setdefault("_step_times_ns", [])
            hist.append(now - prev)
            if len(hist) >= 64:
                hist.sort()
                board.publish_fast_step(hist[len(hist) // 10])
                del hist[:]
        self._step_started_ns = now if started else 0
        board.publish_step(now if started else 0)

    # ---------------------------------------------------------------------------- loop
    def step(self) -> bool:
        if self.enable_overlap:
            return self.step_overlap()
        t0 = time.perf_counter()
        recv = self.recv_requests()
        self.process_input_requests(recv)
        batch = self.get_next_batch_to_run()
        if batch is None:
            self.flush_stream_output()
            self._share_step(0)
            return bool(recv)
        t1 = time.perf_counter()
        self._share_step(len(batch.reqs))
        self._publish_step(True)       # (plain loop: the GPU is idle when a step is launched)
        logits_output, next_token_ids = self.run_batch(batch)  # asynchronous: one hipGraph launch
        batch.output_ids = next_token_ids
        self.flush_stream_output()     # tokens of the previous step: pickle + send while the GPU works
        ids = next_token_ids.tolist()  # the only device sync of a decode step
        self._publish_step(False)
        t2 = time.perf_counter()
        self.process_batch_result_decode(batch, ids, self.extract_logprobs(logits_output))
        t3 = time.perf_counter()
        st = self.stats
        st["t_schedule_s"] = st.get("t_schedule_s", 0.0) + (t1 - t0)
        st["t_forward_s"] = st.get("t_forward_s", 0.0) + (t2 - t1)
        st["t_output_s"] = st.get("t_output_s", 0.0) + (t3 - t2)
        return True

    # ---------------------------------------------------------------------------- overlap schedule
    def step_overlap(self) -> bool:
        """One iteration of the overlapped loop (scheduler.py event_loop_overlap, tp_worker_overlap_thread.py
        :142-235).  The next step's input ids are the previous step's sampled ids ON THE DEVICE, so step k + 1 is
        scheduled and launched while step k runs; the tokens of step k come to the host through a pinned buffer
        and an event and are processed (finish checks, KV release, streaming) while step k + 1 runs.  A request
        that ends at step k is therefore part of step k + 1 once more: that token is dropped and only the KV slot
        taken for it is released then (scheduler.py:1437-1442 "free the one delayed token")."""
        t0 = time.perf_counter()
        recv = self.recv_requests()
        self.process_input_requests(recv)
        rb = self.running_batch
        # penalties are a function of a request's generated tokens (sampling_batch_info.py): while any running
        # request has them, each step waits for the previous one's ids, i.e. the loop degrades to the plain one
        needs_history = any(r.sampling_params.needs_penalties for r in rb.reqs)
        if self._pending is not None and not rb.is_empty() and \
                (needs_history or not rb.check_decode_mem(2) or self.forced_retractions(rb)):
            # a retraction re-sends origin_input_ids + output_ids: every sampled token must be on the host first
            self._drain_pending()
        batch = self.get_next_batch_to_run()
        if batch is None:
            had = self._pending is not None
            self._drain_pending()
            self.flush_stream_output()
            self._share_step(0)
            return bool(recv) or had
        t1 = time.perf_counter()
        self._share_step(len(batch.reqs))
        if self._pending is None:
            self._publish_step(True)   # nothing in flight: this step begins now (otherwise: when its predecessor ends)
        logits_output, next_token_ids = self.run_batch(batch)  # asynchronous: one hipGraph launch
        batch.output_ids = next_token_ids
        bs = len(batch.reqs)
        if torch.device(self.device).type == "cuda":
            if self._pinned_ids is None or self._pinned_ids[0].numel() < bs:
                n = max(bs, 2 * self.max_running_requests)
                self._pinned_ids = [torch.empty(n, dtype=torch.int64, pin_memory=True) for _ in range(2)]
            self._pinned_flip ^= 1
            host_ids = self._pinned_ids[self._pinned_flip][:bs]
            host_ids.copy_(next_token_ids, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:  # CPU runs of the protocol tests: same control flow, nothing to wait for
            host_ids, ev = next_token_ids.to(torch.int64).clone(), None
        prev, self._pending = self._pending, (list(batch.reqs), batch.out_cache_loc, host_ids, ev, logits_output)
        t2 = time.perf_counter()
        if prev is not None:
            self._process_pending(prev)
        self.flush_stream_output()
        t3 = time.perf_counter()
        st = self.stats
        st["t_schedule_s"] = st.get("t_schedule_s", 0.0) + (t1 - t0)
        st["t_forward_s"] = st.get("t_forward_s", 0.0) + (t2 - t1)
        st["t_output_s"] = st.get("t_output_s", 0.0) + (t3 - t2)
        return True

    def _drain_pending(self):
        if self._pending is not None:
            prev, self._pending = self._pending, None
            self._process_pending(prev)

    def _process_pending(self, pending):
        reqs, out_cache_loc, host_ids, ev, logits_output = pending
        if ev is not None:
            self._wait_servicing(ev)           # step k and its copy are done; step k + 1 keeps the GPU busy
            # step k + 1 (if one was launched behind k) began on the GPU just now; otherwise nothing is in flight
            self._publish_step(self._pending is not None)
            ttft_trace.mark("d_step_done", [str(len(reqs))])
        ids = host_ids.tolist()
        logprobs = self.extract_logprobs(logits_output)
        alloc = self.token_to_kv_pool_allocator
        alloc.free_group_begin()
        live = []
        for i, (req, tok) in enumerate(zip(reqs, ids)):
            if req.finished():
                # it ended one step earlier and ran once more: drop the token, release the slot of that step
                alloc.free(req.kv_slots)
                req.kv_slots = []
                continue
            req.output_ids.append(int(tok))
            self._record_logprob(req, i, logprobs)
            req.check_finished()
            if req.finished():
                self.tree_cache.cache_finished_req(req)
            live.append(req)
        alloc.free_group_end()
        self.stats["decode_steps"] += 1
        self.stats["decode_tokens"] += len(live)
        self.stream_output(live, defer=False)
        self.last_progress = time.monotonic()

    # what may be handled while a step is in flight and the previous one is not accounted yet: enqueueing a request
    # (host list only), an admission (KV / request slots come from the host allocator; the reply is built from host
    # state) and a prefill result (merged into the running batch that the NEXT schedule reads).  Everything else -- aborts,
    # statistics resets, cache flushes, shutdown -- touches the running batch or the loop's own state and waits for the
    # loop top, in arrival order.
    _SERVICED_IN_WAIT = (TokenizedGenerateReqInput, GetNextPrefillBatchInput, BatchProcessPrefillResultReq)

    def _wait_servicing(self, ev):
        """Wait for a decode step on the GPU, answering the prefill instance meanwhile.  The loop comes round once
        per decode step (6-8 ms); an admission request or a prefill result that arrives just after a launch would
        otherwise sit in the socket for the rest of the step -- half a step on average, on the TTFT path twice
        (admission, then first token).  Only the three message kinds of _SERVICED_IN_WAIT are handled here (host work
        plus asynchronous copies queued behind the step in flight, which owns copies of its inputs); the rest is kept
        for the loop top.  The wait never nests: nothing it dispatches may wait for a step itself.
        With tensor parallelism every rank must see the same messages at the same point: plain wait."""
        if self.tp_size > 1 or self.recv_from_tokenizer is None:
            ev.synchronize()
            return
        assert not self._in_wait, "_wait_servicing re-entered: a message handled inside the wait waited for a step"
        self._in_wait = True
        try:
            while not ev.query():
                recv = self.recv_requests()
                if recv:
                    now = [r for r in recv if isinstance(r, self._SERVICED_IN_WAIT)]
                    self._deferred_input.extend(r for r in recv if not isinstance(r, self._SERVICED_IN_WAIT))
                    if now:
                        self.process_input_requests(now)
                else:
                    time.sleep(30e-6)
        finally:
            self._in_wait = False

    def recv_requests(self) -> list:
        """The loop top also gets what a wait set aside (in arrival order, before anything newer)."""
        recv = super().recv_requests()
        if self._deferred_input and not self._in_wait:
            recv = self._deferred_input + recv
            self._deferred_input = []
        return recv

    def _synthetic_step(self) -> bool:
        """One captured decode step (the graph closest to 32 requests, inputs as the capture left them: dummy slot 0)
        for nobody: the weight stream of a busy decode instance, for the prefill instance to tune next to."""
        gr = getattr(self.model_runner, "graph_runner", None)
        if gr is None or self.tp_size > 1 or not gr.graphs or time.monotonic() > self._synthetic_load_until:
            self._synthetic_load_until = 0.0
            return False
        key = min(gr.graphs, key=lambda cb: (cb[0] != self.model_runner.num_cus_owned, abs(cb[1] - 32)))
        gr.graphs[key].replay()
        torch.cuda.current_stream().synchronize()
        self.last_progress = time.monotonic()
        return True

    def event_loop_normal(self):
        while not self._shutdown:
            if not self.step():
                if self._synthetic_load_until and self.running_batch.is_empty() and self._synthetic_step():
                    continue
                self.check_watchdog()
                self.idle_sleep()
