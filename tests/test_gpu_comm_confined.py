"""Collectives of an instance on its CU share stay on the share (round-4 verdict item 4a).

In `--cu-mask-mode dynamic` the processes are unmasked; a prefill instance under tensor parallelism therefore has to keep
its all-reduces inside its share itself -- the reference's per-process MPS percentage confines NCCL implicitly
(entrypoints/engine.py:591-593, 632-634).  model_executor/cu_share.py gives the communication stream of the overlapped
all-reduce the share's CU mask and routes EVERY all-reduce through the peer-memory kernels (payloads above their limit in
pieces: RCCL would launch on its own unmasked stream).  Here: two TP ranks on the one GPU, each with masked compute and
communication streams; the communicator's CU trace (semipd_ar_set_cu_trace: every block of every collective kernel marks
its hardware CU slot) must stay inside what the placement probe finds for the same mask, and the sums must be right."""
import pytest

from test_gpu_all_reduce import run_world

pytestmark = pytest.mark.gpu


def test_every_collective_kernel_of_a_masked_instance_runs_inside_its_mask(device):
    reports = run_world(2, extra_env={"HSA_CU_MASK": "0:0-255"}, worker="comm_confined_worker.py", timeout=300)
    assert sorted(r["rank"] for r in reports) == [0, 1]
    for r in reports:
        assert r["cases"] == 5 and not r["bad"], r
        # 96 logical CUs per rank: the probe must see about that many hardware slots, the collectives some of them, none outside
        assert 80 <= r["allowed_slots"] <= 100, r
        assert r["traced_slots"] >= 8 and r["outside_the_mask"] == [], r
        # both overlapped reduces ran as peer-memory kernels on the communication stream
        assert r["overlap_stats"]["overlapped_reduces"] == 2 and r["overlap_stats"]["overlapped_reduces_peer_memory_kernel"] == 2, r
