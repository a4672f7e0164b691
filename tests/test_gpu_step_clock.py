"""Decode-step deadline gate (csrc/step_clock.hip, semi_pd/step_clock.py) on the GPU.

 * the kernels: a gate passes at once while no step is in flight or the step is younger than the deadline; it holds its
   stream while an overdue step is in flight and lets go when the stamp is cleared; a stamp nobody clears (a dead decode
   instance) is waited for max_wait at most;
 * the engine: with a deadline so short that every decode step counts as overdue, the prefill instance's gates hold at
   layer boundaries -- across processes, on the slot imported through the IPC info -- and the tokens are still the
   oracle's (the gate orders streams, it computes nothing).
The reference has no counterpart (static MPS percentages, semi_pd/utils.py:10-11): no reference test exists."""
import time

import pytest
import torch

from oracle.model import OracleLlama
from test_gpu_engine import check_against_oracle, make_prompts, server_args, tiny_llama

pytestmark = pytest.mark.gpu


def _spin(ms, stream, out):
    """Keep `stream` busy for about `ms` milliseconds (the placement probe's spin loop; 100 MHz wall clock x 1e5 per ms is
    not what clock64 counts -- the probe spins on the shader clock, ~2.4 GHz)."""
    from semi_pd_amd import _lib
    with torch.cuda.stream(stream):
        _lib.check(_lib.load().semipd_probe_cu_placement(out.data_ptr(), 1, int(ms * 2.0e6), stream.cuda_stream), "probe")


def test_gate_passes_holds_and_times_out(device):
    from semi_pd_amd.semi_pd import step_clock as SC
    clock = SC.StepClock.create(device)
    try:
        peer = SC.StepClock(device, clock.slot, owner=False)     # same process, same slot: the gating side
        peer.set_deadline_ms(1.0)
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        out = torch.zeros(8, dtype=torch.int32, device=device)
        # nothing in flight: the gate returns at once
        with torch.cuda.stream(b):
            peer.gate()
        torch.cuda.synchronize()
        assert peer.stats() == {"gates": 1, "holds": 0, "held_ms": 0.0, "timeouts": 0}
        # a step in flight for ~30 ms on stream a; the gate arrives 5 ms into it (older than the 1 ms deadline) and must hold
        # stream b until the stamp is cleared
        with torch.cuda.stream(a):
            clock.mark(True)
        _spin(30, a, out)
        with torch.cuda.stream(a):
            clock.mark(False)
            end_a = torch.cuda.Event(enable_timing=True)
            end_a.record()
        time.sleep(0.005)
        t0 = time.perf_counter()
        with torch.cuda.stream(b):
            peer.gate()
            end_b = torch.cuda.Event(enable_timing=True)
            end_b.record()
        end_b.synchronize()
        waited = (time.perf_counter() - t0) * 1e3
        assert end_a.query(), "the gate let go before the step's end stamp"
        st = peer.stats()
        assert st["gates"] == 2 and st["holds"] == 1 and st["timeouts"] == 0
        assert 5.0 < st["held_ms"] < 60.0 and waited > 5.0, (st, waited)
        # a young step is not waited for
        peer.set_deadline_ms(10_000.0)
        with torch.cuda.stream(a):
            clock.mark(True)
        torch.cuda.synchronize()
        with torch.cuda.stream(b):
            peer.gate()
        torch.cuda.synchronize()
        assert peer.stats()["holds"] == 1
        # a stamp that is never cleared: max_wait bounds the hold
        peer.set_deadline_ms(0.01)
        time.sleep(0.002)
        t0 = time.perf_counter()
        with torch.cuda.stream(b):
            peer.gate()
        torch.cuda.synchronize()
        waited = (time.perf_counter() - t0) * 1e3
        st = peer.stats()
        assert st["holds"] == 2 and st["timeouts"] == 1 and 0.8 * SC.MAX_WAIT_MS < waited < 4 * SC.MAX_WAIT_MS, (st, waited)
        with torch.cuda.stream(a):
            clock.mark(False)
        torch.cuda.synchronize()
    finally:
        clock.close()


def test_the_gate_is_capturable_and_the_stamp_travels_in_the_decode_graph(device):
    from semi_pd_amd.semi_pd import step_clock as SC
    clock = SC.StepClock.create(device)
    try:
        peer = SC.StepClock(device, clock.slot, owner=False)
        peer.set_deadline_ms(0.5)
        x = torch.ones(1024, device=device)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            clock.mark(True)
            y = x * 2
            clock.mark(False)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        # slot[0] == 0 (no step in flight), slot[1] == 3 steps begun
        assert clock.peek() == (0, 3)
        assert torch.equal(y, x * 2)
        # a gate captured into a graph: replayed while nothing is in flight it passes
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=cap):
            peer.gate()
            z = x + 1
        torch.cuda.synchronize()
        g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(z, x + 1) and peer.stats()["holds"] == 0 and peer.stats()["gates"] >= 1
    finally:
        clock.close()


def test_semi_pd_with_a_deadline_every_step_misses_still_matches_the_oracle(device):
    from semi_pd_amd.entrypoints.engine import Engine
    from semi_pd_amd.managers.io_struct import SamplingParams
    cfg = tiny_llama()
    prompts = make_prompts(cfg.vocab_size, [5, 37, 128, 1, 64, 90, 17, 33, 200, 150, 11, 75])
    sp = SamplingParams(max_new_tokens=24, ignore_eos=True)
    uni = Engine(server_args(cfg))
    try:
        sd = {k: v.float().cpu() for k, v in uni.model_runner.model.state_dict().items()}
    finally:
        uni.shutdown()
    oracle = OracleLlama(cfg, sd)
    eng = Engine(server_args(cfg, enable_semi_pd=True, cu_mask_mode="dynamic", prefill_cu_percent=88, decode_cu_percent=100,
                             tune_prefill_gemm=False, decode_step_deadline_ms=0.001, chunked_prefill_size=64))
    try:
        got = eng.generate(prompts, sp, timeout=300)
        again = eng.generate(prompts[::-1], sp, timeout=300)
        stats = {s["role"]: s for s in eng.get_stats()}
    finally:
        eng.shutdown()
    check_against_oracle(oracle, prompts, got)
    check_against_oracle(oracle, prompts[::-1], again)
    gate = stats["PREFILL"]["step_gate"]
    # one gate per decoder layer per prefill batch; whether one of them met a decode step in flight is a matter of timing
    # (tiny model: steps of ~1 ms), so only the plumbing is asserted: launches counted, nothing timed out
    assert gate["gates"] >= cfg.num_hidden_layers * stats["PREFILL"]["prefill_batches"] > 0 and gate["timeouts"] == 0
    print("step gate:", gate)
