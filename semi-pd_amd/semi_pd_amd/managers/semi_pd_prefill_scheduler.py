"""SemiPDPrefillScheduler: the prefill instance never allocates.  It proposes request ids to the
decode instance, receives (rids, req_pool_indices, prefix_lens, extend_input_lens), reads its
out_cache_loc from the *shared* req_to_token table, runs the prefill forward and ships the sampled
token ids back.  Reference: managers/semi_pd_prefill_scheduler.py:40-176."""
from __future__ import annotations

import logging
import os
import time
from typing import List, Optional

import torch

from semi_pd_amd.distributed import broadcast_pyobj
from semi_pd_amd.managers.io_struct import (AbortReq, BatchProcessPrefillResultReq, GetNextPrefillBatchInput,
                                            GetNextPrefillBatchOutput, TokenizedGenerateReqInput)
from semi_pd_amd.managers.schedule_batch import ScheduleBatch
from semi_pd_amd.managers.scheduler import SchedulerBase
from semi_pd_amd.semi_pd import ttft_trace
from semi_pd_amd.semi_pd.utils import InstanceRole

logger = logging.getLogger(__name__)


class BatchTimeModel:
    """GPU seconds of a prefill batch of n tokens, d = a + b n, fitted to the recent batches with exponential forgetting
    (weighted least squares over running sums).  While the sizes seen so far do not spread (every batch one 1024-token
    request), the ratio d / n of the weighted means stands in; before any batch, a guess of 20 us per token."""

    def __init__(self, decay: float = 0.85):
        self.decay = decay
        self.s1 = self.sn = self.sd = self.snn = self.snd = 0.0
        self._seen = 0        # samples offered
        self._outliers = 0    # consecutive samples set aside as outliers

    def update(self, n: int, d: float) -> None:
        if n <= 0 or d <= 0:
            return
        self._seen += 1
        if self._seen == 1:
            return            # a process's first batch pays for lazily loaded code objects and first-touch allocations
        if self.s1 > 0 and d > 3.0 * self.predict(n) and self._outliers < 2:
            self._outliers += 1   # a stall (another first-time shape): not the law; three in a row ARE the law
            return
        self._outliers = 0
        k = self.decay
        self.s1, self.sn, self.sd = k * self.s1 + 1.0, k * self.sn + n, k * self.sd + d
        self.snn, self.snd = k * self.snn + float(n) * n, k * self.snd + n * d

    def predict(self, n: int) -> float:
        if self.s1 <= 0:
            return 20e-6 * n
        det = self.s1 * self.snn - self.sn * self.sn
        if det > 1e-3 * self.s1 * self.snn:                    # the sizes spread enough for a slope
            b = (self.s1 * self.snd - self.sn * self.sd) / det
            a = (self.sd - b * self.sn) / self.s1
            if b > 0 and a >= 0:
                return a + b * n
        return self.sd / self.sn * n


class SemiPDPrefillScheduler(SchedulerBase):
    def __init__(self, server_args, model_runner, tp_rank, recv_socket, send_to_d_instance, bridge_socket,
                 send_stats_to=None):
        # send_stats_to: only StatsReq answers go there; tokens are streamed by the decode instance
        super().__init__(server_args, model_runner, tp_rank, recv_socket, send_stats_to, InstanceRole.PREFILL)
        # pipelined loop (on unless --disable-overlap-schedule): the next batch is launched behind the running one
        # before the running one's ids are read, so the GPU does not idle between prefill batches
        self.enable_overlap = not getattr(server_args, "disable_overlap_schedule", False)
        self._inflight = None             # (batch, ids on the host side, event, logits_output, t_launch)
        # late binding (on by default, one GPU per instance; SEMIPD_PREFILL_LATE_BIND=0: the early proposal of round 2):
        # the batch that follows a running one is put together as late as its launch still lands behind the running
        # one without a gap -- requests that arrive while a batch runs make the NEXT batch instead of the one after it
        self.late_bind = (self.enable_overlap and server_args.tp_size == 1 and recv_socket is not None
                          and torch.device(model_runner.device).type == "cuda"
                          and os.environ.get("SEMIPD_PREFILL_LATE_BIND", "1") != "0")
        self.lead_s = float(os.environ.get("SEMIPD_PREFILL_LEAD_MS", "4.0")) * 1e-3
        self._batch_time = BatchTimeModel()   # GPU seconds of a batch as a function of its tokens
        self._gpu_free_at = 0.0           # when the last finished batch left the GPU (perf_counter)
        self._watch = None                # the running batch whose end the layer hooks of the next launch look for
        self._in_wait = False
        self._deferred_input: list = []   # messages a wait set aside for the loop top
        if self.late_bind:
            self._install_layer_hooks(model_runner)
            pacer = getattr(model_runner, "step_pacer", None)
            if pacer is not None and os.environ.get("SEMIPD_PACER_NO_WAIT_HOOK") != "1":   # (A / B knob)
                # the paced forward spends most of its time waiting inside the layer hooks: look for the end of the batch
                # that ran before it there as well (semi_pd/step_pacer.py)
                pacer.while_waiting = lambda: self._between_layers(None, None)
        self._share_inflight = 0          # batches launched and not finished (published on the share board)
        # prompt tokens waiting from which a batch takes every CU whatever the decode instance does (0 = never)
        self.backlog_full_tokens = int(getattr(server_args, "prefill_backlog_full_tokens", 0) or 0)
        self._proposal_in_flight = False  # a GetNextPrefillBatchInput whose reply has not been read yet
        self._aborted: set = set()        # rids the client gave up on (see abort_request)
        self.chunked_rid: Optional[str] = None
        self.send_to_d_instance = send_to_d_instance  # PUSH -> D's input socket (rank 0 only)
        self.bridge_socket = bridge_socket            # PULL <- D's replies (rank 0 only)

    def add_to_waiting_queue(self, req):
        ttft_trace.mark("p_recv", [req.rid])
        if req.is_retracted:
            # retracted requests jump the queue, like the decode side does (semi_pd_decode_scheduler.py:137)
            self.waiting_queue.insert(0, req)
        else:
            self.waiting_queue.append(req)

    def abort_request(self, recv_req):
        """The decode instance is the authority: it drops a queued request, or marks an admitted one.  Here
        an aborted rid stays proposable until a reply shows which of the two happened: a reply that
        contains it means it was admitted (run it; D finishes it after the merge), a reply without it means
        D no longer knows it (drop it)."""
        self._aborted.add(recv_req.rid)

    def _purge_aborted(self, admitted_rids):
        if self._aborted:
            keep = set(admitted_rids)
            self.waiting_queue = [r for r in self.waiting_queue if r.rid not in self._aborted or r.rid in keep]
            self._aborted &= {r.rid for r in self.waiting_queue} | keep

    def to_extend_batch(self, resp: GetNextPrefillBatchOutput) -> ScheduleBatch:
        """semi_pd_prefill_scheduler.py:74-118."""
        can_run_list = [r for r in self.waiting_queue if r.rid in resp.rids]
        can_run_list.sort(key=lambda r: resp.rids.index(r.rid))
        if self.chunked_rid != resp.chunked_rid:
            # the last chunked request has finished prefilling: drop it from the waiting queue
            new_waiting_queue = []
            for r in self.waiting_queue:
                if r.rid == self.chunked_rid:
                    continue
                if r.rid in resp.rids and r.rid != resp.chunked_rid:
                    continue
                new_waiting_queue.append(r)
            self.waiting_queue = new_waiting_queue
            self.chunked_rid = resp.chunked_rid
        else:
            self.waiting_queue = [r for r in self.waiting_queue
                                  if r.rid not in resp.rids or r.rid == resp.chunked_rid]
        table = self.req_to_token_pool.req_to_token
        for i, r in enumerate(can_run_list):
            assert r.rid == resp.rids[i]
            r.extend_input_len = resp.extend_input_lens[i]
            pre_len = resp.prefix_lens[i]
            r.prefix_indices = table[resp.req_pool_indices[i], :pre_len].to(torch.int64)
            r.fill_ids = r.origin_input_ids[: pre_len + r.extend_input_len]
        batch = ScheduleBatch.init_new(can_run_list, self.req_to_token_pool, self.token_to_kv_pool_allocator,
                                       self.tree_cache, self.device)
        batch.prepare_for_extend(pre_allocated_req_pool_indices=resp.req_pool_indices,
                                 pre_allocated_slots=getattr(resp, "extend_slots", None))
        return batch

    def _propose(self) -> bool:
        """Rank 0: send the decode instance the rids we would like to prefill next
        (semi_pd_prefill_scheduler.py:120-133).  Returns False when nothing is waiting."""
        n_prefill_tokens = 0
        candidates: List[str] = []
        for r in self.waiting_queue:
            if n_prefill_tokens > self.chunked_prefill_size:
                break
            n_prefill_tokens += len(r.origin_input_ids)
            candidates.append(r.rid)
        if not candidates:
            return False
        ttft_trace.mark("p_propose", candidates)
        self.send_to_d_instance.send_pyobj(GetNextPrefillBatchInput(rids=candidates))
        return True

    def request_next_batch_early(self):
        """Ask for the NEXT batch while the current one is still running on the GPU: the decode instance
        answers between two of its steps (up to one step, ~9 ms, later), which the reference pays as idle
        time of the prefill instance before every batch.  The reply is picked up by the next
        get_next_batch_to_run; the decode instance keeps its scheduled prefill batches in FIFO order, so
        results and admissions stay matched."""
        # not while a chunked request is in flight: the decode instance refuses to admit before the
        # previous chunk's result has arrived (semi_pd_decode_scheduler.py:310-320)
        if self.tp_rank == 0 and not self._proposal_in_flight and self.waiting_queue and self.chunked_rid is None:
            self._proposal_in_flight = self._propose()

    def get_next_batch_to_run(self, block: bool = True) -> Optional[ScheduleBatch]:
        """semi_pd_prefill_scheduler.py:120-157.  block=False (a batch is running on the GPU): take the decode
        instance's reply only if it is already there."""
        resp = None
        if self.tp_rank == 0 and (self._proposal_in_flight or self.waiting_queue):
            if not self._proposal_in_flight and (block or self.chunked_rid is None):
                self._proposal_in_flight = self._propose()
            if self._proposal_in_flight and block:
                # the reference blocks forever here (semi_pd_prefill_scheduler.py:134); we bound the wait
                t0 = time.perf_counter()
                resp = self.bridge_socket.recv_pyobj(timeout=self.server_args.watchdog_timeout)
                self.stats["t_wait_admission_s"] = self.stats.get("t_wait_admission_s", 0.0) + time.perf_counter() - t0
            elif self._proposal_in_flight:
                from semi_pd_amd.managers.transport import NOTHING
                got = self.bridge_socket.recv_pyobj_nowait()
                resp = None if got is NOTHING else got
            if resp is not None:
                self._proposal_in_flight = False
                assert isinstance(resp, GetNextPrefillBatchOutput), f"unexpected bridge message {type(resp)}"
        if self.tp_size > 1:
            resp = broadcast_pyobj([resp], self.tp_rank, self.tp_cpu_group, src=0)[0]
        if resp is not None:
            self._purge_aborted(list(resp.rids) + ([resp.chunked_rid] if resp.chunked_rid else []))
        if resp and len(resp.rids) > 0:
            ttft_trace.mark("p_admitted", resp.rids)
            return self.to_extend_batch(resp)
        if resp is not None:
            self._rejected = True  # the decode instance has no room right now: back off a little
        return None

    def process_batch_result_prefill(self, batch: ScheduleBatch, next_token_ids: torch.Tensor, logits_output=None):
        """semi_pd_prefill_scheduler.py:159-173.  `.tolist()` synchronises the stream, so every KV row
        this batch wrote is in HBM before the decode instance hears about it (SURVEY §3.2 hazard)."""
        ids = next_token_ids.tolist()
        self.stats["prefill_batches"] += 1
        self.stats["prefill_tokens"] += batch.extend_num_tokens
        if self.tp_rank == 0:
            self.send_to_d_instance.send_pyobj(BatchProcessPrefillResultReq(
                next_token_ids=ids, next_token_logprobs=self.extract_logprobs(logits_output)))

    # ---------------------------------------------------------------------------- late binding
    # what may be handled while a batch is on the GPU: enqueueing a request (host list) and noting an abort (a set);
    # everything else (statistics, flush, shutdown) waits for the loop top, in arrival order
    _SERVICED_IN_WAIT = (TokenizedGenerateReqInput, AbortReq)

    def _install_layer_hooks(self, model_runner):
        """Launching a prefill batch keeps the host busy for ~5 ms (eager launches, layer by layer).  When that batch
        is queued BEHIND a running one, the running one usually ends inside those 5 ms: the hook (called before every
        decoder layer) then sends its first tokens at once instead of after the last launch."""
        import torch.nn as nn
        model = getattr(model_runner, "model", None)
        if model is None:
            return
        for m in model.modules():
            if isinstance(m, nn.ModuleList):
                for layer in m:
                    layer.register_forward_pre_hook(self._between_layers)

    def _between_layers(self, module, args):
        w = self._watch
        # (not with logprobs to fetch: that copy would queue behind the layers already launched)
        if w is not None and getattr(w[3], "next_token_logprobs", None) is None and w[2].query():
            self._watch = None
            self.stats["results_sent_from_layer_hook"] = self.stats.get("results_sent_from_layer_hook", 0) + 1
            self._finish(w)

    def _predicted_end(self, inflight) -> float:
        batch, _, _, _, t0 = inflight
        return max(t0, self._gpu_free_at) + self._batch_time.predict(batch.extend_num_tokens)

    def _wait_launching_next(self, prev):
        """Wait for `prev` on the GPU; take new requests meanwhile; `lead_s` before its predicted end ask the decode
        instance for the next batch (everything that has arrived until then) and launch it behind `prev`.  `prev` is
        finished -- its ids sent -- by the first layer hook of that launch that finds it done, or here."""
        ev = prev[2]
        launch_at = self._predicted_end(prev) - self.lead_s
        launched = False
        self._in_wait = True
        try:
            while not ev.query():
                recv = self.recv_requests()
                if recv:
                    now = [r for r in recv if isinstance(r, self._SERVICED_IN_WAIT)]
                    self._deferred_input.extend(r for r in recv if not isinstance(r, self._SERVICED_IN_WAIT))
                    self.process_input_requests(now)
                if (not launched and self.chunked_rid is None and (self._proposal_in_flight or self.waiting_queue)
                        and time.perf_counter() >= launch_at):
                    self._rejected = False
                    nxt = self.get_next_batch_to_run(block=False)
                    if nxt is not None:
                        launched = True
                        self.stats["late_bound_launches"] = self.stats.get("late_bound_launches", 0) + 1
                        self._watch = prev
                        try:
                            self._launch(nxt)      # self._inflight = nxt; a layer hook may finish prev on the way
                        finally:
                            watched, self._watch = self._watch, None
                        if watched is None:
                            return                 # a hook has finished prev
                    elif self._rejected:           # the decode instance has no room: ask again a little later
                        launch_at = time.perf_counter() + 2e-3
                    continue                       # (proposal in flight: poll for the reply without the sleep)
                time.sleep(50e-6)
        finally:
            self._in_wait = False
        if not launched:
            self._inflight = None
        self._finish(prev)

    def recv_requests(self) -> list:
        """The loop top also gets what a wait set aside (in arrival order, before anything newer)."""
        recv = super().recv_requests()
        if self._deferred_input and not self._in_wait:
            recv = self._deferred_input + recv
            self._deferred_input = []
        return recv

    def step(self) -> bool:
        self.process_input_requests(self.recv_requests())
        if not self.enable_overlap:
            batch = self.get_next_batch_to_run()
            if batch is None:
                return False
            self._launch(batch)
            self._finish()
            return True
        # Pipelined: while batch i runs, the reply for batch i + 1 (asked for right after the launch of i) is
        # usually there long before i is done; batch i + 1 is then queued behind i, and only after that the ids of
        # i are awaited and sent.  Results reach the decode instance in launch order, which is the order of its
        # scheduled_prefill_batches.
        if self._inflight is None:
            batch = self.get_next_batch_to_run()
            if batch is None:
                return False
            self._launch(batch)
            return True
        if self.late_bind and self._inflight[2] is not None:
            self._wait_launching_next(self._inflight)
            return True
        nxt = self.get_next_batch_to_run(block=False) if self.chunked_rid is None else None
        prev = self._inflight
        if nxt is not None:
            self._launch(nxt)
        else:
            self._inflight = None
        self._finish(prev)
        return True

    def _share_launch(self, batch: ScheduleBatch):
        """--cu-mask-mode dynamic (model_executor/cu_share.py): a batch runs on every CU while the decode instance has
        nothing in flight, or while this instance's backlog says the GPU is overloaded anyway (more than
        `backlog_full_tokens` prompt tokens waiting: throughput first, the decode instance's steps then share their CUs
        with it as in the reference's overlapping MPS shares); on the prefill share otherwise."""
        share = getattr(self.model_runner, "cu_share", None)
        if share is None:
            return
        self._share_inflight += 1
        share.publish(self._share_inflight)
        backlog = sum(len(r.origin_input_ids) for r in self.waiting_queue)
        from semi_pd_amd.model_executor.cu_share import FULL
        # (one decision per forward: under tensor parallelism rank 0's, CuShare.decide)
        overloaded = bool(self.backlog_full_tokens and backlog >= self.backlog_full_tokens)
        name = share.decide(FULL if overloaded else None)
        share.activate(name)
        pacer = getattr(self.model_runner, "step_pacer", None)
        if pacer is not None:
            pacer.hold_enabled = not overloaded      # throughput first: no holds for decode steps that are all overdue
        self.stats["batches_on_" + name] = self.stats.get("batches_on_" + name, 0) + 1

    def _share_done(self):
        share = getattr(self.model_runner, "cu_share", None)
        if share is not None:
            self._share_inflight = max(0, self._share_inflight - 1)
            share.publish(self._share_inflight)

    def _launch(self, batch: ScheduleBatch):
        t0 = time.perf_counter()
        self._share_launch(batch)
        logits_output, next_token_ids = self.run_batch(batch)  # asynchronous launches
        if torch.device(self.device).type == "cuda" and self.enable_overlap:
            host_ids = torch.empty(next_token_ids.numel(), dtype=torch.int64, pin_memory=True)
            host_ids.copy_(next_token_ids, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host_ids, ev = next_token_ids, None
        self._inflight = (batch, host_ids, ev, logits_output, t0)
        ttft_trace.mark("p_launched", [r.rid for r in batch.reqs])
        if not self.late_bind and os.environ.get("SEMIPD_EARLY_PROPOSE", "1") != "0":
            self.request_next_batch_early()

    def _finish(self, inflight=None):
        if inflight is None:
            inflight, self._inflight = self._inflight, None
        batch, host_ids, ev, logits_output, t0 = inflight
        if ev is not None:
            ev.synchronize()  # the batch and the copy of its ids are done: every KV row it wrote is in HBM
        t_done = time.perf_counter()
        self._share_done()
        ttft_trace.mark("p_done", [r.rid for r in batch.reqs])
        if ev is not None:
            # this batch had the GPU from its launch, or from the end of the batch it was queued behind
            owned = t_done - max(t0, self._gpu_free_at)
            self._batch_time.update(batch.extend_num_tokens, owned)
            # (t_forward_s below runs from the launch: for a batch queued behind a running one it includes the rest of THAT
            #  batch -- up to `lead_s` -- which is not GPU time of this one; bench.py reports both)
            self.stats["t_gpu_owned_s"] = self.stats.get("t_gpu_owned_s", 0.0) + owned
        self._gpu_free_at = t_done
        self.process_batch_result_prefill(batch, host_ids, logits_output)
        self.stats["t_forward_s"] = self.stats.get("t_forward_s", 0.0) + time.perf_counter() - t0
        self.stats["prefill_reqs"] = self.stats.get("prefill_reqs", 0) + len(batch.reqs)
        self.last_progress = time.monotonic()

    def event_loop_normal(self):
        import time
        while not self._shutdown:
            self._rejected = False
            if not self.step():
                if self._rejected:
                    time.sleep(0.0005)
                else:
                    self.idle_sleep()
