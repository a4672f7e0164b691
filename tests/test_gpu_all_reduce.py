"""GPU test of the peer-memory TP all-reduce against oracle.ops.all_reduce_sum.

The box has one GPU, so the ranks are `world` processes on cuda:0: the regions are exchanged through
hipIpc handles exactly as between GPUs, the kernels poll each other's flags the same way, only the wire
is HBM instead of xGMI.  HSA_CU_MASK gives every rank its own CUs, so the spin-waits cannot starve the
rank they wait for.  The reference's own test is test/srt/test_custom_allreduce.py (eager + graph,
random sizes, against NCCL); here the expected bits come from the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def run_world(world, extra_env=None, timeout=180, worker="ar_worker.py"):
    import torch
    num_cus = torch.cuda.get_device_properties(0).multi_processor_count
    share = num_cus // world // 8 * 8
    port = 23000 + (os.getpid() * 7 + world) % 4000
    procs = []
    for r in range(world):
        env = dict(os.environ, HSA_CU_MASK=f"0:{r * share}-{(r + 1) * share - 1}")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(r), str(world), str(port)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    reports, logs = [], []
    try:
        for p in procs:
            try:
                out, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                out, _ = p.communicate()
                raise AssertionError(f"rank timed out after {timeout}s:\n{out[-2000:]}") from None
            logs.append(out)
            line = [l for l in out.splitlines() if l.startswith("AR_REPORT ")]
            assert p.returncode == 0 and line, f"rank exited with {p.returncode}:\n{out[-3000:]}"
            reports.append(json.loads(line[-1][len("AR_REPORT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return reports


@pytest.mark.parametrize("world", [2, 4, 8])
def test_all_reduce_matches_oracle_bit_for_bit(device, world):
    reports = run_world(world)
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        assert r["cases"] >= 3 * 14 + 48 + 1
        assert not r["bad"], r["bad"][:5]
    print(f"\nworld {world}: 512 KB bf16 all-reduce {max(r['us_per_call_512KB'] for r in reports):.1f} us per call "
          f"(ranks share one GPU), 48 mixed calls {max(r['mixed_48_calls_ms'] for r in reports):.2f} ms")


def test_all_reduce_few_blocks_and_forced_two_stage(device):
    """Same cases with 3 blocks per launch (every block loops over a long chunk) and with the one-stage
    kernel switched off (two-stage even for 16 bytes: empty slices, empty sub-chunks)."""
    for r in run_world(4, {"SEMIPD_AR_MAX_BLOCKS": "3", "SEMIPD_AR_ONE_SHOT_BELOW": "0"}):
        assert not r["bad"], r["bad"][:5]


@pytest.mark.parametrize("phase", ["export", "map", "selftest"])
def test_a_rank_that_cannot_set_up_makes_every_rank_fall_back(device, phase):
    """Start-up is a sequence of votes on the CPU group: a failure on one rank (injected here) disables the
    peer-memory path on all ranks — the engine then reduces through RCCL — and nothing hangs or raises."""
    reports = run_world(4, {"SEMIPD_AR_TEST_FAIL": f"{phase}:2"}, timeout=120)
    assert len(reports) == 4 and all(r.get("disabled") for r in reports)


@pytest.mark.parametrize("world", [2, 4])
def test_reference_op_set_by_name(device, world):
    """The reference's ROCm custom all-reduce ops (sgl-kernel/csrc/torch_extension_rocm.cc:25-55; python wrappers
    sgl_kernel/allreduce.py:5-52) exist by name in semi_pd_amd/sgl_kernel_allreduce.py and, called in the order the
    reference's CustomAllreduce calls them, reduce to the oracle's bits -- eagerly, through all_reduce_unreg with a
    registered buffer, and from a captured graph whose (empty) buffer list is registered afterwards."""
    reports = run_world(world, worker="ar_opset_worker.py")
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        assert r["cases"] == 3 * 4 * 2 + 2 and not r["bad"], r["bad"][:5]
