"""CPU test of the P<->D protocol — host logic only, no kernels.

The prefill and decode schedulers run in one process against in-memory sockets and a *shared*
req_to_token tensor (standing in for the hipIpcMemHandle mapping).  A toy "model" stores token ids
in a fake KV array at out_cache_loc and derives the next token from the tokens it gathers through
req_to_token — so any mistake in slot allocation, in the shared table, in chunked prefill or in the
retract path changes the generated ids.  Expected ids come from the full token history alone, and
must also equal what the unified scheduler produces (the invariant that pins Semi-PD)."""
import collections
from types import SimpleNamespace

import pytest
import torch

from semi_pd_amd.managers.io_struct import SamplingParams, TokenizedGenerateReqInput
from semi_pd_amd.managers.schedule_batch import ScheduleBatch
from semi_pd_amd.managers.scheduler import Scheduler
from semi_pd_amd.managers.semi_pd_decode_scheduler import SemiPDDecodeScheduler
from semi_pd_amd.managers.semi_pd_prefill_scheduler import SemiPDPrefillScheduler
from semi_pd_amd.managers.transport import NOTHING
from semi_pd_amd.mem_cache.memory_pool import ReqToTokenPool, TokenToKVPoolAllocator
from semi_pd_amd.server_args import ServerArgs

VOCAB = 997


def toy_next(history):
    return (sum((i + 1) * t for i, t in enumerate(history)) * 31 + len(history)) % VOCAB


def expected(prompt, n):
    h = list(prompt)
    out = []
    for _ in range(n):
        t = toy_next(h)
        out.append(t)
        h.append(t)
    return out


class Q:
    """In-memory PUSH/PULL pair."""

    def __init__(self, pump=None):
        self.q = collections.deque()
        self.pump = pump

    def send_pyobj(self, obj):
        self.q.append(obj)

    def recv_pyobj_nowait(self):
        return self.q.popleft() if self.q else NOTHING

    def recv_pyobj(self, timeout=None):
        for _ in range(10000):
            if self.q:
                return self.q.popleft()
            self.pump()  # let the peer scheduler run (stands in for the other process)
        raise TimeoutError


class FakeWorker:
    def __init__(self, runner, kv_store):
        self.runner, self.kv = runner, kv_store

    def forward_batch_generation(self, mwb):
        table = self.runner.req_to_token_pool.req_to_token
        self.kv[mwb.out_cache_loc.long()] = mwb.input_ids
        ids = []
        for rp, sl in zip(mwb.req_pool_indices.tolist(), mwb.seq_lens.tolist()):
            slots = table[rp, :sl].long()
            assert (slots > 0).all(), "a sequence points at the dummy slot 0"
            assert len(set(slots.tolist())) == sl, "two tokens of one sequence share a KV slot"
            ids.append(toy_next(self.kv[slots].tolist()))
        return None, torch.tensor(ids, dtype=torch.int32)


def make_runner(shared=None, size=4000, max_reqs=32, ctx=512):
    if shared is None:
        r2t = ReqToTokenPool(max_reqs + 1, ctx + 4, "cpu")
    else:
        r2t = ReqToTokenPool(max_reqs + 1, ctx + 4, "cpu", bypass_create_buffers=True)
        r2t.req_to_token = shared.req_to_token_pool.req_to_token  # the IPC mapping
    alloc = TokenToKVPoolAllocator(size, torch.bfloat16, "cpu", None)
    return SimpleNamespace(device=torch.device("cpu"), req_to_token_pool=r2t,
                           token_to_kv_pool_allocator=alloc, max_total_num_tokens=size)


def args(**kw):
    base = dict(model_config=None, context_length=512, max_running_requests=32, max_total_tokens=4000,
                chunked_prefill_size=8192, enable_semi_pd=True, watchdog_timeout=5.0)
    base.update(kw)
    return ServerArgs(**base)


def run_semi_pd(prompts, max_new, sa, size=4000, force_retract=0, interleave=True, sampling_kw=None,
                worker_cls=None):
    d_runner = make_runner(size=size)
    p_runner = make_runner(shared=d_runner, size=size)
    kv = torch.zeros(size + 1, dtype=torch.int64)
    d_in, p_in, out = Q(), Q(), Q()
    holder = {}
    bridge = Q(pump=lambda: holder["d"].step())
    d = SemiPDDecodeScheduler(sa, d_runner, 0, d_in, out, bridge, p_in)
    p = SemiPDPrefillScheduler(sa, p_runner, 0, p_in, d_in, bridge)
    holder["d"] = d
    d.tp_worker, p.tp_worker = (worker_cls or FakeWorker)(d_runner, kv), FakeWorker(p_runner, kv)
    if force_retract:
        d.forced_retractions = lambda batch: force_retract if batch.batch_size() > 4 else 0
    got = {}
    new_of = max_new if isinstance(max_new, (list, tuple)) else [max_new] * len(prompts)  # per request or one for all
    reqs = [TokenizedGenerateReqInput(f"r{i}", None, list(pr), SamplingParams(
        max_new_tokens=new_of[i], ignore_eos=True, **((sampling_kw or (lambda i: {}))(i)))) for i, pr in enumerate(prompts)]
    pending = collections.deque(reqs)
    for it in range(20000):
        if pending and (not interleave or it % 3 == 0):
            r = pending.popleft()
            d_in.send_pyobj(r)   # D first, then P (tokenizer_manager.py:149-160)
            p_in.send_pyobj(r)
        p.step()
        d.step()
        while out.q:
            o = out.q.popleft()
            for rid, toks in zip(o.rids, o.output_ids):
                got.setdefault(rid, []).extend(toks)
        if not pending and len(got) == len(reqs) and all(len(got[f"r{i}"]) >= n for i, n in enumerate(new_of)):
            break
    for _ in range(3):  # the event loops keep running: the overlapped decode loop still holds its last (surplus) step
        p.step()
        d.step()
    # nothing leaks: every KV slot and request slot is back in the decode instance's pools
    assert d_runner.token_to_kv_pool_allocator.available_size() == size
    assert d_runner.req_to_token_pool.available_size() == d_runner.req_to_token_pool.size
    assert p_runner.token_to_kv_pool_allocator.available_size() == size, "the prefill instance allocated KV"
    assert not d.scheduled_prefill_batches and not d.waiting_queue and not p.waiting_queue
    return [got[f"r{i}"] for i in range(len(reqs))], d, p


def run_unified(prompts, max_new, sa, size=4000):
    runner = make_runner(size=size)
    kv = torch.zeros(size + 1, dtype=torch.int64)
    inbox, out = Q(), Q()
    s = Scheduler(sa, runner, 0, inbox, out)
    s.tp_worker = FakeWorker(runner, kv)
    new_of = max_new if isinstance(max_new, (list, tuple)) else [max_new] * len(prompts)
    for i, pr in enumerate(prompts):
        inbox.send_pyobj(TokenizedGenerateReqInput(f"r{i}", None, list(pr),
                                                   SamplingParams(max_new_tokens=new_of[i], ignore_eos=True)))
    got = {}
    for _ in range(20000):
        if not s.step() and len(got) == len(prompts) and all(len(got[f"r{i}"]) >= n for i, n in enumerate(new_of)):
            break
        while out.q:
            o = out.q.popleft()
            for rid, toks in zip(o.rids, o.output_ids):
                got.setdefault(rid, []).extend(toks)
    assert runner.token_to_kv_pool_allocator.available_size() == size
    return [got[f"r{i}"] for i in range(len(prompts))]


def prompts_of(lens, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(1, VOCAB, (n,), generator=g).tolist() for n in lens]


def test_semi_pd_equals_unified_equals_history_rule():
    prompts = prompts_of([5, 40, 128, 1, 77, 33, 64, 12])
    want = [expected(p, 9) for p in prompts]
    assert run_unified(prompts, 9, args(enable_semi_pd=False)) == want
    got, d, p = run_semi_pd(prompts, 9, args())
    assert got == want
    assert p.stats["prefill_tokens"] == sum(len(x) for x in prompts)
    assert d.stats["decode_tokens"] == 8 * len(prompts)


def test_chunked_prefill_across_p_and_d():
    prompts = prompts_of([150, 20, 200, 9, 77, 130], seed=1)
    want = [expected(p, 6) for p in prompts]
    assert run_unified(prompts, 6, args(enable_semi_pd=False, chunked_prefill_size=64)) == want
    got, d, p = run_semi_pd(prompts, 6, args(chunked_prefill_size=64))
    assert got == want
    assert p.stats["prefill_batches"] > len(prompts)  # the long prompts really were split


def test_retracted_requests_are_re_prefilled():
    prompts = prompts_of([30, 8, 51, 17, 23, 40, 12, 9], seed=2)
    want = [expected(p, 12) for p in prompts]
    got, d, p = run_semi_pd(prompts, 12, args(), force_retract=2, interleave=False)
    assert got == want
    assert p.stats["prefill_tokens"] > sum(len(x) for x in prompts), "no request was re-prefilled"


def test_memory_pressure_retract_and_admission_control():
    # pool of 300 slots: admission control (PrefillAdder budget) and OOM retraction must both hold
    prompts = prompts_of([60, 70, 50, 40, 65, 30], seed=3)
    want = [expected(p, 20) for p in prompts]
    got, d, p = run_semi_pd(prompts, 20, args(max_total_tokens=300), size=300, interleave=False)
    assert got == want


def test_single_token_requests_finish_at_prefill():
    prompts = prompts_of([10, 3, 25], seed=4)
    got, d, p = run_semi_pd(prompts, 1, args())
    assert got == [expected(p_, 1) for p_ in prompts]
    assert d.stats["decode_tokens"] == 0


def test_one_token_requests_do_not_leave_the_full_flag_set():
    """A prefill batch whose requests all end at their first token (max_new_tokens = 1) merges nothing into the
    running batch; when the admission that produced it had set batch_is_full (pool exhausted, or max_running_requests
    reached by the in-flight prefill batches) nothing clears the flag again while the running batch stays empty, and
    the decode instance would refuse every later admission with all KV slots free (the unified scheduler resets the
    flag when the last batch shrinks, scheduler.py:1043-1047)."""
    for seed in (4, 6, 20):   # seeds that stalled before the flag was reset (4, 6: max_running_requests; 20: pool)
        g = torch.Generator().manual_seed(seed)
        lens = torch.randint(5, 60, (40,), generator=g).tolist()
        prompts = prompts_of(lens, seed=seed)
        new = [1 if torch.rand(1, generator=g) < 0.7 else int(torch.randint(2, 9, (1,), generator=g))
               for _ in range(len(prompts))]
        want = [expected(p, n) for p, n in zip(prompts, new)]
        for kw in (dict(max_total_tokens=200, max_running_requests=32),
                   dict(max_total_tokens=4000, max_running_requests=4)):
            size = kw["max_total_tokens"]
            for plain in (False, True):
                got, d, _ = run_semi_pd(prompts, new, args(disable_overlap_schedule=plain, **kw), size=size,
                                        interleave=False)
                assert [g_[:n] for g_, n in zip(got, new)] == want
            assert run_unified(prompts, new, args(enable_semi_pd=False, **kw), size=size) == want


def test_plain_and_overlapped_decode_loops_agree():
    """The decode instance's loop is overlapped by default (the tests above run it): step k + 1 is scheduled
    before the tokens of step k are processed, a finished request runs one surplus step whose token is dropped and
    whose slot is released later, a retraction drains the pending step first.  --disable-overlap-schedule gives the
    plain loop; both produce the history-rule tokens and leak nothing (run_semi_pd asserts the pools)."""
    prompts = prompts_of([5, 40, 17, 1, 33, 8, 21, 2, 64, 11])
    for kw in (dict(), dict(chunked_prefill_size=16), dict(force_retract=2)):
        force = kw.pop("force_retract", 0)
        outs = {}
        for plain in (False, True):
            got, d, _ = run_semi_pd(prompts, 9, args(disable_overlap_schedule=plain, **kw), force_retract=force)
            assert d.enable_overlap == (not plain)
            outs[plain] = got
        assert outs[False] == outs[True] == [expected(p, 9) for p in prompts]


def test_penalties_see_every_generated_token_in_the_overlapped_loop():
    """Penalties are a function of a request's generated tokens (sampling/penaltylib).  The overlapped loop launches
    step k + 1 before the token of step k has reached the host, so while a running request asks for penalties each
    step waits for the previous one's ids: the penalty entries handed to the worker must then count every token
    generated so far, for every such request, at every step.  Requests without penalties in the same batch are
    unaffected, and the tokens stay the history-rule tokens (the fake worker does not apply the penalties)."""
    prompts = prompts_of([5, 40, 17, 9, 33, 8])
    plen = {}
    seen = {"steps": 0, "rows": 0}

    class CheckingWorker(FakeWorker):
        def forward_batch_generation(self, mwb):
            info = mwb.sampling_info
            rows = [i for i, r in enumerate(self.batch_reqs()) if r.sampling_params.frequency_penalty]
            if rows:
                assert info.has_penalties
                seen["steps"] += 1
                for i in rows:
                    r = self.batch_reqs()[i]
                    counted = info.pen_vals[info.pen_rows == i].sum().item()
                    generated = int(mwb.seq_lens[i]) - len(r.origin_input_ids)
                    assert counted == generated == len(r.output_ids), (r.rid, counted, generated, len(r.output_ids))
                    seen["rows"] += 1
            return super().forward_batch_generation(mwb)

    holder = {}
    CheckingWorker.batch_reqs = lambda self: holder["d"].running_batch.reqs
    for kw in (dict(), dict(force_retract=2)):
        seen.update(steps=0, rows=0)
        # run_semi_pd gives the worker no handle on the scheduler: take it from the first get_next_batch_to_run
        orig = SemiPDDecodeScheduler.get_next_batch_to_run

        def spy(self_):
            holder["d"] = self_
            return orig(self_)

        SemiPDDecodeScheduler.get_next_batch_to_run = spy
        try:
            got, d, _ = run_semi_pd(prompts, 9, args(), worker_cls=CheckingWorker, force_retract=kw.get("force_retract", 0),
                                    sampling_kw=lambda i: {"frequency_penalty": 1.0} if i % 2 == 0 else {})
        finally:
            SemiPDDecodeScheduler.get_next_batch_to_run = orig
        assert d.enable_overlap and got == [expected(p, 9) for p in prompts]
        assert seen["steps"] >= 8 and seen["rows"] >= 3 * 8


def test_wait_for_a_step_services_only_the_pd_messages_and_defers_the_rest():
    """While the decode instance waits for a step it answers the prefill instance (admissions, prefill results) and
    enqueues new requests; anything that touches the running batch or the loop's own state -- an abort, a statistics
    reset -- is kept, in arrival order, for the top of the loop.  The wait does not nest."""
    from semi_pd_amd.managers.io_struct import AbortReq, GetNextPrefillBatchInput, StatsReq
    sa = args()
    d_runner = make_runner()
    d_in, out, bridge, p_in = Q(), Q(), Q(), Q()
    d = SemiPDDecodeScheduler(sa, d_runner, 0, d_in, out, bridge, p_in)

    class Ev:   # an event that completes after a few polls; messages arrive in between
        def __init__(self):
            self.polls = 0

        def query(self):
            self.polls += 1
            if self.polls == 2:
                d_in.send_pyobj(TokenizedGenerateReqInput("w0", None, [1, 2, 3], SamplingParams(max_new_tokens=4, ignore_eos=True)))
                d_in.send_pyobj(AbortReq("nobody"))
                d_in.send_pyobj(StatsReq(reset=True))
                d_in.send_pyobj(GetNextPrefillBatchInput(["w0"]))
            return self.polls > 4

        def synchronize(self):
            raise AssertionError("the servicing wait polls, it does not block")

    d.stats["decode_steps"] = 7
    d._wait_servicing(Ev())
    # the request was enqueued and admitted inside the wait (the reply to P is on the bridge), the other two were kept
    assert not d.waiting_queue and len(d.scheduled_prefill_batches) == 1
    assert bridge.q, "no admission reply was sent during the wait"
    assert [type(m).__name__ for m in d._deferred_input] == ["AbortReq", "StatsReq"]
    assert d.stats["decode_steps"] == 7, "a statistics reset ran inside the wait"
    # the loop top hands them on, before anything newer, and the wait cannot be entered from inside itself
    d_in.send_pyobj(StatsReq(reset=False))
    got = d.recv_requests()
    assert [type(m).__name__ for m in got] == ["AbortReq", "StatsReq", "StatsReq"] and not d._deferred_input
    d._in_wait = True
    with pytest.raises(AssertionError):
        d._wait_servicing(Ev())


def _late_binding_rig(finish_in_hook):
    """P and D in one process; P's batches carry a fake event (the CPU path has none) so that the late-binding wait of
    the GPU path runs: the event of the first batch completes only after `done["first"]` is set."""
    import time
    sa = args()
    d_runner = make_runner()
    p_runner = make_runner(shared=d_runner)
    kv = torch.zeros(4001, dtype=torch.int64)
    d_in, p_in, out, bridge = Q(), Q(), Q(), Q()
    d = SemiPDDecodeScheduler(sa, d_runner, 0, d_in, out, bridge, p_in)
    p = SemiPDPrefillScheduler(sa, p_runner, 0, p_in, d_in, bridge)
    bridge.pump = d.step
    state = {"polls": 0, "first_done": False, "order": []}

    class HookedWorker(FakeWorker):   # a model whose layers call the scheduler's hook, like the real one's
        def forward_batch_generation(self, mwb):
            if finish_in_hook and len(state["order"]) == 1:
                state["first_done"] = True        # the running batch ends while the next one is being launched
            p._between_layers(None, ())
            return super().forward_batch_generation(mwb)

    d.tp_worker, p.tp_worker = FakeWorker(d_runner, kv), HookedWorker(p_runner, kv)
    p.late_bind, p.lead_s = True, 0.0

    def send(rid, prompt):
        r = TokenizedGenerateReqInput(rid, None, list(prompt), SamplingParams(max_new_tokens=3, ignore_eos=True))
        d_in.send_pyobj(r)
        p_in.send_pyobj(r)

    class Ev:
        def __init__(self, first):
            self.first = first

        def query(self):
            d.step()                              # the decode instance is another process: it runs meanwhile
            if not self.first:
                return True
            state["polls"] += 1
            if state["polls"] == 2:
                send("b", prompts[1])
            if state["polls"] == 5:
                send("c", prompts[2])
                from semi_pd_amd.managers.io_struct import StatsReq
                p_in.send_pyobj(StatsReq(reset=True))
            if not finish_in_hook and state["polls"] > 40:
                state["first_done"] = True
            return state["first_done"]

        def synchronize(self):
            assert self.query(), "a batch was waited for with a blocking call before it was done"

    launch = p._launch

    def launch_with_event(batch):
        launch(batch)
        b, ids, _, lo, t0 = p._inflight
        p._inflight = (b, ids, Ev(first=not state["order"]), lo, t0)
        state["order"].append([r.rid for r in b.reqs])

    p._launch = launch_with_event
    handle_stats = p.handle_stats
    state["stats_handled_in_wait"] = []
    p.handle_stats = lambda req: (state["stats_handled_in_wait"].append(p._in_wait), handle_stats(req))
    # the next batch is asked for 30 ms after the wait begins: "b" and "c" have both arrived by then
    p._predicted_end = lambda prev: time.perf_counter() + 0.03
    prompts = prompts_of([40, 33, 21], seed=5)
    send("a", prompts[0])
    got = {}
    for _ in range(400):
        p.step()
        d.step()
        while out.q:
            o = out.q.popleft()
            for rid, toks in zip(o.rids, o.output_ids):
                got.setdefault(rid, []).extend(toks)
        if len(got) == 3 and all(len(v) >= 3 for v in got.values()):
            break
    return p, d, state, got, prompts


@pytest.mark.parametrize("finish_in_hook", [False, True])
def test_late_binding_puts_arrivals_during_a_batch_into_the_next_one(finish_in_hook):
    """While a prefill batch runs, new requests are taken from the socket; shortly before the batch's predicted end
    ONE next batch is admitted with everything that has arrived and launched behind it; the running batch's ids go
    out as soon as it is done -- from the layer hook of the launch in progress, or from the wait -- and before the
    next batch's (the decode instance matches results to admissions in order).  Other messages wait for the loop top."""
    p, d, state, got, prompts = _late_binding_rig(finish_in_hook)
    assert state["order"] == [["a"], ["b", "c"]], state["order"]
    assert state["stats_handled_in_wait"] == [False] and not p._deferred_input
    for rid, pr in zip("abc", prompts):
        assert got[rid][:3] == expected(pr, 3), rid
    assert not d.scheduled_prefill_batches and not p.waiting_queue and p._inflight is None


def test_synthetic_load_request_is_bounded_and_needs_a_captured_graph():
    """SyntheticLoadReq (prefill instance -> decode instance while the prefill GEMMs are timed): on arms a deadline, off
    clears it; without captured graphs (this CPU rig), with tensor parallelism or past the deadline nothing is replayed
    and the flag drops, so an idle decode loop never spins for a prefill instance that went away."""
    import time
    from semi_pd_amd.managers.io_struct import SyntheticLoadReq
    d_in, out, bridge, p_in = Q(), Q(), Q(), Q()
    d = SemiPDDecodeScheduler(args(), make_runner(), 0, d_in, out, bridge, p_in)
    d_in.send_pyobj(SyntheticLoadReq(on=True))
    d.step()
    assert d._synthetic_load_until > time.monotonic() + 60
    assert d._synthetic_step() is False and d._synthetic_load_until == 0.0     # no graph runner here
    replays = []
    # graphs are keyed (CUs of the stream they were captured for, batch size): the set of the current stream is taken
    d.model_runner.num_cus_owned = 96
    d.model_runner.graph_runner = SimpleNamespace(graphs={(96, 8): SimpleNamespace(replay=lambda: replays.append(8)),
                                                          (96, 32): SimpleNamespace(replay=lambda: replays.append(32)),
                                                          (256, 32): SimpleNamespace(replay=lambda: replays.append(-32)),
                                                          (96, 64): SimpleNamespace(replay=lambda: replays.append(64))})
    d_in.send_pyobj(SyntheticLoadReq(on=True))
    d.step()
    sync = torch.cuda.current_stream
    try:
        torch.cuda.current_stream = lambda *a, **k: SimpleNamespace(synchronize=lambda: None)
        assert d._synthetic_step() is True and replays == [32]
        d._synthetic_load_until = time.monotonic() - 1.0                         # the deadline has passed
        assert d._synthetic_step() is False and replays == [32]
        d_in.send_pyobj(SyntheticLoadReq(on=True))
        d_in.send_pyobj(SyntheticLoadReq(on=False))
        d.step()
        assert d._synthetic_load_until == 0.0
    finally:
        torch.cuda.current_stream = sync


def test_batch_time_model_recovers_an_affine_law_and_degrades_gracefully():
    """The prefill scheduler's predictor of a batch's GPU time (late binding asks for the next batch a lead before the
    predicted end): a guess before any sample, the ratio of the means while all batches have one size, the weighted
    least-squares line once sizes spread, recent batches weighing more, nonsense fits (negative slope) refused."""
    from semi_pd_amd.managers.semi_pd_prefill_scheduler import BatchTimeModel
    m = BatchTimeModel()
    assert m.predict(1000) == pytest.approx(0.02)
    m.update(1024, 95e-3)                          # a process's first batch (lazy code-object loads): not a sample
    assert m.predict(1000) == pytest.approx(0.02)
    for _ in range(5):
        m.update(1024, 22.3e-3)
    m.update(1024, 90e-3)                          # one stall: set aside ...
    assert m.predict(1024) == pytest.approx(22.3e-3, rel=1e-6)
    m.update(1024, 22.3e-3)
    assert m.predict(1024) == pytest.approx(22.3e-3, rel=1e-6) and m.predict(2048) == pytest.approx(44.6e-3, rel=1e-6)
    for n in (2048, 1024, 3072, 1024, 2048, 1024):
        m.update(n, 4e-3 + 17.87e-6 * n)           # T(n) = 4 ms + 18.3 ms per 1024 tokens
    assert m.predict(2048) == pytest.approx(4e-3 + 17.87e-6 * 2048, rel=0.03)
    assert m.predict(4096) == pytest.approx(4e-3 + 17.87e-6 * 4096, rel=0.05)
    for _ in range(40):                            # the law changes (another share): the old samples fade
        for n in (1024, 2048):
            m.update(n, 2e-3 + 10e-6 * n)
    assert m.predict(3072) == pytest.approx(2e-3 + 10e-6 * 3072, rel=0.02)
    slow = BatchTimeModel()
    slow.update(1024, 1.0)
    slow.update(1024, 10e-3)
    for _ in range(3):                             # ... three in a row are the new law
        slow.update(1024, 50e-3)
    assert slow.predict(1024) > 20e-3
    bad = BatchTimeModel()
    bad.update(500, 1.0)                           # (first sample of a process: skipped)
    bad.update(1000, 10e-3)
    bad.update(2000, 5e-3)                         # a shorter time for more tokens: no line through that
    assert bad.predict(3000) == pytest.approx((10e-3 * 0.85 + 5e-3) / (1000 * 0.85 + 2000) * 3000)
    bad.update(0, 1.0), bad.update(10, -1.0)       # ignored


def test_late_binding_survives_a_cold_first_batch(monkeypatch):
    """The whole pipelined loop against a simulated GPU and clock: batches take 1 ms + 5 us per token, the process's FIRST
    batch 60x that (lazy code-object loads).  With the real predictor (BatchTimeModel) the slow first batch must not keep
    the following bursts from being launched behind running batches (it did, when it was a sample: the estimate stayed
    several times too long and every batch ended before its successor was asked for)."""
    from semi_pd_amd.managers import semi_pd_prefill_scheduler as mod

    class Clock:                                   # stands in for the `time` module inside the scheduler
        def __init__(self):
            self.t = 100.0

        def perf_counter(self):
            self.t += 2e-6
            return self.t

        monotonic = time = perf_counter

        def sleep(self, dt):
            self.t += dt

    clock = Clock()
    monkeypatch.setattr(mod, "time", clock)
    sa = args()
    d_runner = make_runner()
    p_runner = make_runner(shared=d_runner)
    kv = torch.zeros(4001, dtype=torch.int64)
    d_in, p_in, out, bridge = Q(), Q(), Q(), Q()
    d = SemiPDDecodeScheduler(sa, d_runner, 0, d_in, out, bridge, p_in)
    p = SemiPDPrefillScheduler(sa, p_runner, 0, p_in, d_in, bridge)
    bridge.pump = d.step
    d.tp_worker, p.tp_worker = FakeWorker(d_runner, kv), FakeWorker(p_runner, kv)
    p.late_bind, p.lead_s = True, 0.4e-3
    gpu = {"free_at": 0.0, "batches": 0}

    class Ev:
        def __init__(self, end):
            self.end = end

        def query(self):
            d.step()
            return clock.t >= self.end

        def synchronize(self):
            clock.t = max(clock.t, self.end)

    launch = p._launch

    def launch_on_simulated_gpu(batch):
        launch(batch)
        clock.t += 0.3e-3                          # the host's launches
        b, ids, _, lo, t0 = p._inflight
        dur = 1e-3 + 5e-6 * b.extend_num_tokens
        if gpu["batches"] == 0:
            dur *= 60
        gpu["batches"] += 1
        gpu["free_at"] = max(gpu["free_at"], t0) + dur
        p._inflight = (b, ids, Ev(gpu["free_at"]), lo, t0)

    p._launch = launch_on_simulated_gpu
    prompts = prompts_of([200, 180, 190, 170, 160, 200], seed=9)
    got, n_sent = {}, 0
    for burst in range(4):
        pending = collections.deque((f"b{burst}r{i}", pr) for i, pr in enumerate(prompts))
        next_send = clock.t
        for _ in range(20000):
            if pending and clock.t >= next_send:
                rid, pr = pending.popleft()
                r = TokenizedGenerateReqInput(rid, None, list(pr), SamplingParams(max_new_tokens=2, ignore_eos=True))
                d_in.send_pyobj(r), p_in.send_pyobj(r)
                n_sent += 1
                next_send = clock.t + 0.25e-3      # arrivals 0.25 ms apart: they fall into running batches
            if not p.step():
                clock.sleep(50e-6)
            d.step()
            while out.q:
                o = out.q.popleft()
                for rid, toks in zip(o.rids, o.output_ids):
                    got.setdefault(rid, []).extend(toks)
            if not pending and len(got) == n_sent and all(len(v) >= 2 for v in got.values()) and p._inflight is None:
                break
    for burst in range(4):
        for i, pr in enumerate(prompts):
            assert got[f"b{burst}r{i}"][:2] == expected(pr, 2)
    print("late-bound launches:", p.stats.get("late_bound_launches", 0), "of", p.stats["prefill_batches"], "batches")
    assert p.stats.get("late_bound_launches", 0) >= 8, p.stats
    # the cold batch was not a sample: a 200-token batch is predicted at about its true 2 ms, not at tens of ms
    assert p._batch_time.predict(200) < 4e-3


def _simulated_serving(seed, monkeypatch, n_reqs=24, size=1500, max_reqs=6, chunk=8192, abort_some=False):
    """P and D against a simulated GPU and clock (see the test above), with randomised arrival gaps, prompt lengths, batch
    durations (+-30 % noise, occasional 5x stalls) and a decode instance small enough to refuse admissions.  Returns
    (expected tokens per request, received tokens per request, schedulers)."""
    import random
    from semi_pd_amd.managers import semi_pd_prefill_scheduler as mod
    from semi_pd_amd.managers.io_struct import AbortReq
    rnd = random.Random(seed)

    class Clock:
        def __init__(self):
            self.t = 50.0

        def perf_counter(self):
            self.t += 2e-6
            return self.t

        monotonic = time = perf_counter

        def sleep(self, dt):
            self.t += dt

    clock = Clock()
    monkeypatch.setattr(mod, "time", clock)
    sa = args(max_running_requests=max_reqs, chunked_prefill_size=chunk, max_total_tokens=size)
    d_runner = make_runner(size=size, max_reqs=max_reqs)
    p_runner = make_runner(shared=d_runner, size=size, max_reqs=max_reqs)
    kv = torch.zeros(size + 1, dtype=torch.int64)
    d_in, p_in, out, bridge = Q(), Q(), Q(), Q()
    d = SemiPDDecodeScheduler(sa, d_runner, 0, d_in, out, bridge, p_in)
    p = SemiPDPrefillScheduler(sa, p_runner, 0, p_in, d_in, bridge)
    bridge.pump = d.step

    class LayeredWorker(FakeWorker):              # a model of four layers: host time passes between them, the hook runs
        def forward_batch_generation(self, mwb):
            for _ in range(4):
                clock.t += rnd.uniform(0.02e-3, 0.15e-3)
                p._between_layers(None, ())
            return super().forward_batch_generation(mwb)

    d.tp_worker, p.tp_worker = FakeWorker(d_runner, kv), LayeredWorker(p_runner, kv)
    p.late_bind, p.lead_s = True, rnd.choice([0.1e-3, 0.4e-3, 1.5e-3])
    gpu = {"free_at": 0.0}

    class Ev:
        def __init__(self, end):
            self.end = end

        def query(self):
            d.step()
            return clock.t >= self.end

        def synchronize(self):
            clock.t = max(clock.t, self.end)

    launch = p._launch

    def launch_on_simulated_gpu(batch):
        launch(batch)
        b, ids, _, lo, t0 = p._inflight
        dur = (0.5e-3 + 5e-6 * b.extend_num_tokens) * rnd.uniform(0.7, 1.3) * (5.0 if rnd.random() < 0.08 else 1.0)
        gpu["free_at"] = max(gpu["free_at"], t0) + dur
        p._inflight = (b, ids, Ev(gpu["free_at"]), lo, t0)

    p._launch = launch_on_simulated_gpu
    lens = [rnd.randint(3, 120) for _ in range(n_reqs)]
    prompts = prompts_of(lens, seed=seed)
    new_of = [rnd.randint(1, 6) for _ in range(n_reqs)]
    aborted = set()
    pending = collections.deque(range(n_reqs))
    next_send, got, sent = clock.t, {}, []
    for _ in range(200000):
        if pending and clock.t >= next_send:
            i = pending.popleft()
            r = TokenizedGenerateReqInput(f"r{i}", None, list(prompts[i]), SamplingParams(max_new_tokens=new_of[i], ignore_eos=True))
            d_in.send_pyobj(r), p_in.send_pyobj(r)
            sent.append(i)
            if abort_some and rnd.random() < 0.15:
                aborted.add(i)
                d_in.send_pyobj(AbortReq(f"r{i}")), p_in.send_pyobj(AbortReq(f"r{i}"))
            next_send = clock.t + rnd.choice([0.0, 0.05e-3, 0.3e-3, 2e-3])
        if not p.step():
            clock.sleep(50e-6)
        d.step()
        while out.q:
            o = out.q.popleft()
            for rid, toks in zip(o.rids, o.output_ids):
                got.setdefault(rid, []).extend(toks)
        live = [i for i in sent if i not in aborted]
        if (not pending and all(len(got.get(f"r{i}", [])) >= new_of[i] for i in live) and p._inflight is None
                and not p.waiting_queue and not d.scheduled_prefill_batches and d.running_batch.is_empty()):
            break
    else:
        raise AssertionError(f"seed {seed}: the simulated serving run did not finish: waiting {len(p.waiting_queue)}, "
                             f"scheduled {len(d.scheduled_prefill_batches)}, running {len(d.running_batch.reqs)}")
    for _ in range(3):
        p.step(), d.step()
    return prompts, new_of, aborted, got, p, d, d_runner, size


@pytest.mark.parametrize("seed", range(25))
@pytest.mark.parametrize("mode", ["plain", "tight", "chunked", "aborts"])
def test_late_binding_randomised_serving_keeps_every_invariant(seed, mode, monkeypatch):
    """Random arrivals, lengths and batch durations through the late-binding loop (simulated GPU and clock): every request
    gets exactly its tokens (= the history rule, so slots, the shared table and result order are right), refused
    admissions and chunked prompts do not wedge the pipeline, aborted requests end, every KV and request slot returns."""
    kw = {"plain": {}, "tight": dict(size=700, max_reqs=3), "chunked": dict(chunk=64),
          "aborts": dict(abort_some=True)}[mode]
    prompts, new_of, aborted, got, p, d, d_runner, size = _simulated_serving(seed * 7 + 1, monkeypatch, **kw)
    for i, (pr, n) in enumerate(zip(prompts, new_of)):
        if i in aborted:
            assert got.get(f"r{i}", []) == expected(pr, n)[: len(got.get(f"r{i}", []))]
            continue
        assert got[f"r{i}"] == expected(pr, n), (mode, seed, i)
    assert d_runner.token_to_kv_pool_allocator.available_size() == size
    assert d_runner.req_to_token_pool.available_size() == d_runner.req_to_token_pool.size
    assert not p._deferred_input and not p._aborted or mode == "aborts"
